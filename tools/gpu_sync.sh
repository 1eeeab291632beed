#!/bin/bash
# device entropy decoder without restart markers: parity tests, then end-to-end rates and a kernel trace
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x -k device_entropy > gpurun_out/pytest_sync.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_sync.log
tail -n 25 gpurun_out/pytest_sync.log
for n in 256 1024; do
  JPGPU_PIPE_TRACE=${TRACE:-} timeout 300 python tools/e2e_bench.py --device-entropy --images $n --no-download 2>&1 | tail -1
done
timeout 300 python tools/e2e_bench.py --device-entropy --images 256 2>&1 | tail -1
for v in "1 12" "3 6" "4 4"; do set -- $v; echo "iters $1 launches $2"; JPGPU_SYNC_ITERS=$1 JPGPU_SYNC_LAUNCHES=$2 timeout 300 python tools/e2e_bench.py --device-entropy --images 256 --no-download 2>&1 | tail -1 | cut -c300-700; done
timeout 300 python tools/e2e_bench.py --device-entropy --images 256 --no-download --file tests/golden/benches/tower.jpg 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/sync_trc -o t -- python $GRAFT_REPO_ROOT/tools/e2e_bench.py --device-entropy --images 256 --no-download --rounds 2 > $GRAFT_REPO_ROOT/gpurun_out/sync_trc.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py gpurun_out/sync_trc 2>&1 | head -30
