#!/bin/bash
# Short form of profile_round.sh: bench lines, the rocprofv3 kernel-trace stats of the bench command and the first PMC pass
# (instruction counts).  The traffic passes are not repeated: nothing about the memory accesses changed.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for wl in 1080p-444 1080p-422 1080p-gray 2160p-420; do
  timeout 120 python bench.py --workload $wl --no-cpu-baseline >> $O/bench_other.jsonl 2>> $O/bench_other.err
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o f -- python $R/bench.py --no-cpu-baseline > $O/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/pmc1 -o p -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/pmc1.log 2>&1
cd $R
python tools/prof_summary.py $O/pmc1 $O/trace > $O/summary.json 2> $O/summary.err
cat $O/bench_default.json
cut -c1-170 $O/bench_other.jsonl
tail -c 1500 $O/summary.json
