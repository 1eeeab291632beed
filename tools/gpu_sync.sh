#!/bin/bash
# device entropy decoder, streams without restart markers (csrc/huff_sync_core.hpp): parity tests, end-to-end rates, kernel trace
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py -m gpu -q -x -k "device_entropy or scan_ranges" > gpurun_out/pytest_sync.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_sync.log
tail -n 5 gpurun_out/pytest_sync.log
: > gpurun_out/e2e_sync.jsonl
for n in 256 1024; do
  timeout 300 python tools/e2e_bench.py --device-entropy --images $n --no-download 2>&1 | tail -1 | tee -a gpurun_out/e2e_sync.jsonl | cut -c1-900
done
JPGPU_PIPE_DEV_SUB=256 timeout 300 python tools/e2e_bench.py --device-entropy --images 1024 --no-download 2>&1 | tail -1 | tee -a gpurun_out/e2e_sync.jsonl | cut -c1-900
timeout 300 python tools/e2e_bench.py --device-entropy --images 256 2>&1 | tail -1 | tee -a gpurun_out/e2e_sync.jsonl | cut -c1-900
timeout 300 python tools/e2e_bench.py --images 256 --no-download 2>&1 | tail -1 | tee -a gpurun_out/e2e_sync.jsonl | cut -c1-900
timeout 300 python tools/e2e_bench.py --device-entropy --images 1024 --no-download --restart-rows 1 2>&1 | tail -1 | tee -a gpurun_out/e2e_sync.jsonl | cut -c1-900
timeout 300 python tools/e2e_bench.py --device-entropy --images 256 --no-download --file tests/golden/benches/tower.jpg 2>&1 | tail -1 | tee -a gpurun_out/e2e_sync.jsonl | cut -c1-900
timeout 300 python tools/e2e_bench.py --device-entropy --images 64 --no-download --file tests/golden/benches/large_image.jpg 2>&1 | tail -1 | tee -a gpurun_out/e2e_sync.jsonl | cut -c1-900
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/sync_trc -o t -- python $GRAFT_REPO_ROOT/tools/e2e_bench.py --device-entropy --images 256 --no-download --rounds 2 > $GRAFT_REPO_ROOT/gpurun_out/sync_trc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py gpurun_out/sync_trc > gpurun_out/sync_trc_summary.json 2>&1
