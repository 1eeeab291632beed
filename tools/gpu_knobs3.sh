#!/bin/bash
# interleaved repeats of the candidates of tools/gpu_knobs2.sh (4,096 files per call; best call and six calls back to back)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/knobs3
out=gpurun_out/knobs3/sweep.txt
: > $out
run() {
  echo -n "$* : " >> $out
  env "$@" timeout 300 python tools/e2e_bench.py --images ${IMAGES:-4096} --device-entropy --no-download --rounds 6 2>&1 \
    | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('best_ms', d['ms']['total_ms'], 'sustained', d['sustained_images_per_s_pixels_left_in_hbm'])" >> $out
}
for rep in $(seq 1 ${REPS:-3}); do
if [ -n "$CANDS" ]; then
  for c in $CANDS; do run $c; done
else
run A=0
run JPGPU_SYNC_BLOCKS=64
run JPGPU_SYNC_TAIL=2
run JPGPU_SYNC_BLOCKS=64 JPGPU_SYNC_TAIL=2
run JPGPU_SYNC_BLOCKS=64 JPGPU_SYNC_TAIL=4
fi
done
sort $out
