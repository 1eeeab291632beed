#!/bin/bash
# GPU pass 2: compiler-probe, smoke, parity tests, bench (fused + generic), rocprof kernel trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe_ashr_pk.hip -o /tmp/probe_ashr 2>/dev/null && /tmp/probe_ashr ) > gpurun_out/probe_ashr_pk.txt 2>&1
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_fused.json 2> gpurun_out/bench_fused.err; echo "bench exit $?" >> gpurun_out/bench_fused.err
timeout 600 python bench.py --steps 10 --warmup 2 --generic --no-cpu-baseline > gpurun_out/bench_generic.json 2> gpurun_out/bench_generic.err
timeout 600 python bench.py --steps 20 --warmup 3 --workload 1080p-444 --no-cpu-baseline > gpurun_out/bench_444.json 2> gpurun_out/bench_444.err
timeout 600 python bench.py --steps 20 --warmup 3 --workload 1080p-gray --no-cpu-baseline > gpurun_out/bench_gray.json 2> gpurun_out/bench_gray.err
timeout 600 python bench.py --steps 20 --warmup 3 --workload 2160p-420 --no-cpu-baseline > gpurun_out/bench_4k.json 2> gpurun_out/bench_4k.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fused -o fused -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_fused.log 2>&1
cd $GRAFT_REPO_ROOT
for f in smoke.log pytest_gpu.log bench_fused.json bench_generic.json bench_444.json bench_gray.json bench_4k.json probe_ashr_pk.txt; do echo "== $f"; tail -n 4 gpurun_out/$f; done
find gpurun_out/prof_fused -name "*stats*" | head
