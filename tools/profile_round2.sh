#!/bin/bash
# Round-2 artefacts for profiles/round2: bench lines of every workload, rocprofv3 kernel-trace stats of the bench command,
# PMC passes (each --pmc set in its own run, kernel trace only), HBM traffic per workload.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof2
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
: > $O/bench_other.jsonl
for wl in 1080p-444 1080p-422 1080p-440 1080p-411 1080p-gray 1080p-cmyk 1080p-444+gray 2160p-420; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-classes >> $O/bench_other.jsonl 2>> $O/bench_other.err
done
JPGPU_420_STRIP=0 timeout 300 python bench.py --no-cpu-baseline --no-classes >> $O/bench_other.jsonl 2>> $O/bench_other.err   # round 1's two-pass 4:2:0 on the same box
timeout 300 python bench.py --workload 2160p-420 --batch 64 --sub-batches 8 --no-cpu-baseline --no-classes >> $O/bench_other.jsonl 2>> $O/bench_other.err  # the N>1 launch-group form on one GPU
timeout 300 python bench.py --generic --steps 50 --warmup 10 --no-cpu-baseline --no-classes >> $O/bench_other.jsonl 2>> $O/bench_other.err
cd /tmp
CMD="python $R/bench.py --no-cpu-baseline --no-classes"
PCMD="python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-classes"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o f -- $CMD > $O/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc1 -o p -- $PCMD > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc2 -o p -- $PCMD > $O/pmc2.log 2>&1
cd $R
python tools/prof_summary.py $O/pmc1 $O/pmc2 $O/trace > $O/kernel_trace_stats_and_pmc.json 2> $O/summary.err
for spec in "1080p-420:fused420:s420_" "1080p-411:fusedgen:fgen_" "1080p-444:fused444:f444_" "1080p-422:fused422:f422_" "1080p-440:fused440:s440_" "1080p-gray:fusedgray:fgray_" "1080p-cmyk:fused444x4:f444_" "2160p-420:fused420:s420_"; do
  IFS=: read wl path pat <<< "$spec"
  cd /tmp
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/tf_$wl -o p -- python $R/bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-classes > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/tw_$wl -o p -- python $R/bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-classes > /dev/null 2>&1
  cd $R/tools && python make_pmc_traffic.py $wl:$path $O/tf_$wl $O/tw_$wl $O/pmc_traffic.json $pat
done
cd /tmp
JPGPU_420_STRIP=0 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/tf_2p -o p -- $PCMD > /dev/null 2>&1
JPGPU_420_STRIP=0 timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/tw_2p -o p -- $PCMD > /dev/null 2>&1
cd $R/tools && python make_pmc_traffic.py 1080p-420:fused420-2pass $O/tf_2p $O/tw_2p $O/pmc_traffic.json f420_
cd $R
find $O/trace -name '*kernel_stats.csv' -exec cp {} $O/rocprofv3_kernel_stats.csv \;
rm -rf $O/trace $O/pmc1 $O/pmc2 $O/tf_* $O/tw_*
cat $O/bench_default.json; cut -c1-40,330-420 $O/bench_other.jsonl | head -0; python - <<PY
import json
for l in open("$O/bench_other.jsonl"):
    l = json.loads(l); print(l["config"]["name"], l["config"]["kernel_path"], l["config"]["sub_batches"], l["ms_per_step"], l["roofline"]["frac"], l["verified_vs_oracle"])
t = json.load(open("$O/pmc_traffic.json"))
for k, v in t.items(): print(k, v["hbm_bytes_per_decode"])
PY
tail -c 1200 $O/kernel_trace_stats_and_pmc.json
