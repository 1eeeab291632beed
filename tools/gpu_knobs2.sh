#!/bin/bash
# After the sync loop got shorter: first-pass tail (eighths of a chunk walked by pass 0) and blocks per chunk, 4,096 and 256 files per call.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/knobs2
out=gpurun_out/knobs2/sweep.txt
: > $out
run() {
  echo "== $*" >> $out
  for n in 4096 256; do
  env "$@" timeout 300 python tools/e2e_bench.py --images $n --device-entropy --no-download --rounds 6 2>&1 \
    | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['images'], 'total_ms', d['ms']['total_ms'], 'images_per_s', d['images_per_s'], 'sustained', d['sustained_images_per_s_pixels_left_in_hbm'])" >> $out
  done
}
run A=0
run JPGPU_SYNC_TAIL=2
run JPGPU_SYNC_TAIL=4
run JPGPU_SYNC_TAIL=5
run JPGPU_SYNC_BLOCKS=32
run JPGPU_SYNC_BLOCKS=64
run JPGPU_SYNC_BLOCKS=96
run A=0
cat $out
