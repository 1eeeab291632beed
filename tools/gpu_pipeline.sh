#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x > gpurun_out/pytest_pipeline.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_pipeline.log
tail -n 12 gpurun_out/pytest_pipeline.log
nproc
for th in 0 128 64; do timeout 600 python tools/e2e_bench.py --threads $th 2>&1 | tail -1; done
timeout 600 python tools/e2e_bench.py --dense 2>&1 | tail -1
timeout 600 python tools/e2e_bench.py --no-download 2>&1 | tail -1
timeout 600 python tools/e2e_bench.py --images 1024 2>&1 | tail -1
timeout 600 python tools/e2e_bench.py --subsampling 4:4:4 2>&1 | tail -1
timeout 600 python tools/e2e_bench.py --progressive --images 128 2>&1 | tail -1
