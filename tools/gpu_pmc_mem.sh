#!/bin/bash
# memory-side counters of the 4:2:0 kernel next to torch.add over the same arenas (bench.py's stream_reference), separate --pmc passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_mem; rm -rf $O; mkdir -p $O
cd /tmp
CMD="python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --class-steps 5"
# (few counters per pass: eight of these at once is "exceeds the capabilities of the hardware", and rocprofv3 then hangs until killed)
i=0
for set in "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ" "TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ" "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL" "TCC_BUSY TCC_CYCLE TCC_TOO_MANY_EA_WRREQS_STALL" "TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ"; do
  i=$((i+1))
  timeout -s KILL 80 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p -- $CMD > $O/p$i.log 2>&1
  echo "pass $i ($set): rc $?"
done
cd $R
python - <<PY
import glob, sqlite3, json
res = {}
for d in ("p1", "p2", "p3", "p4", "p5"):
    for f in glob.glob("$O/" + d + "/*.db"):
        c = sqlite3.connect(f)
        try:
            for k, n, v in c.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
                if "s420_kernel<2" in k or "elementwise" in k.lower() or "add" in k.lower():
                    res.setdefault(k.split("(")[0][:70], {})[n] = round(v, 1)
            for k, cnt, avg in c.execute("select name, count(*), avg(duration) from kernels group by name"):
                kk = k.split("(")[0][:70]
                if kk in res: res[kk].setdefault("_calls_avg_us", [cnt, round(avg / 1e3, 1)])
        except sqlite3.Error as e:
            print("err", e)
json.dump(res, open("$O/pmc_mem.json", "w"), indent=1)
for k, v in res.items():
    print(k)
    for n in sorted(v): print("   ", n, v[n])
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
