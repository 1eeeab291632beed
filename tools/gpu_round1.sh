#!/bin/bash
# First GPU pass: smoke, parity tests, bench (generic path), rocprof kernel trace.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx9|Compute Unit" | head -8 > gpurun_out/rocminfo.txt 2>&1
nproc >> gpurun_out/rocminfo.txt
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_generic.json 2> gpurun_out/bench_generic.err; echo "bench exit $?" >> gpurun_out/bench_generic.err
tail -5 gpurun_out/smoke.log gpurun_out/pytest_gpu.log gpurun_out/bench_generic.json gpurun_out/bench_generic.err
