#!/bin/bash
# quick GPU pass: batch parity tests + benches + kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "batch or smoke or reftest" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
for wl in 1080p-420 1080p-444 1080p-gray 2160p-420; do
  timeout 600 python bench.py --steps 20 --warmup 3 --workload $wl --no-cpu-baseline > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_quick -o q -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_quick.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3,glob
for f in glob.glob('gpurun_out/prof_quick/*.db'):
    c=sqlite3.connect(f)
    for r in c.execute("select name, count(*), avg(duration), min(duration), max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name"):
        if 'jpgpu' in r[0]: print(r)
PY
tail -n 3 gpurun_out/pytest_gpu.log
for wl in 1080p-420 1080p-444 1080p-gray 2160p-420; do python -c "
import json,sys
d=json.load(open('gpurun_out/bench_$wl.json'))
print('$wl', d['config']['kernel_path'], d['value'],'MP/s', d['roofline']['kernel_ms_per_launch'],'ms', d['roofline']['achieved'],'GB/s', d['roofline']['frac'], d['verified_vs_oracle'])
" 2>&1 | tail -1; done
