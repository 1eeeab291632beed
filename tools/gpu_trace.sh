#!/bin/bash
# kernel-trace stats of the default bench command (env knobs pass through): per-kernel durations on stdout
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/trq
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trq -o t -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/trq.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/trq
