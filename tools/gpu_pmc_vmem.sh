#!/bin/bash
# address-translation counters of the 4:2:0 kernel next to torch.add over the same arenas (small sets, hard timeouts)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_vmem; rm -rf $O; mkdir -p $O
cd /tmp
CMD="python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --class-steps 5"
i=0
for set in "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -s KILL 80 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p -- $CMD > $O/p$i.log 2>&1
  echo "pass $i ($set): rc $?"
done
cd $R
python - <<PY
import glob, sqlite3, json
res = {}
for d in ("p1", "p2", "p3", "p4"):
    for f in glob.glob("$O/" + d + "/*.db"):
        c = sqlite3.connect(f)
        try:
            for k, n, v in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
                if "s420_kernel<2" in k or "CUDAFunc" in k:
                    res.setdefault(k.split("(")[0][:70], {})[n] = round(v, 1)
        except sqlite3.Error as e:
            print("err", e)
json.dump(res, open("$O/pmc_vmem.json", "w"), indent=1)
for k, v in res.items():
    print(k)
    for n in sorted(v): print("   ", n, v[n])
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
