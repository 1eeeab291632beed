#!/bin/bash
# Compute streams / hardware queues of jpgpu_pipeline_decode for a call that fills the device (4,096 x 1080p files):
# sub-batch j runs on stream j % streams, so with 16 streams sub-batch 16 waits for sub-batch 0's pixel kernels.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/streams
out=gpurun_out/streams/sweep.txt
: > $out
run() {  # hw queues, streams
  echo "== GPU_MAX_HW_QUEUES=$1 JPGPU_PIPE_STREAMS=$2" >> $out
  GPU_MAX_HW_QUEUES=$1 JPGPU_PIPE_STREAMS=$2 timeout 300 python tools/e2e_bench.py --images 4096 --device-entropy --no-download --rounds 5 2>&1 \
    | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('total_ms', d['ms']['total_ms'], 'images_per_s', d['images_per_s'], 'sustained', d['sustained_images_per_s_pixels_left_in_hbm'])
    else: print(l.rstrip()[:200])" >> $out
}
run 24 16
run 40 32
run 24 16
run 32 24
run 40 32
run 48 32
cat $out
