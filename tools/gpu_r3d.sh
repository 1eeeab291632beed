#!/bin/bash
# round 3, call D: one-sample-row seam transforms (main) against full seam transforms (alt = -DJPGPU_SEAM_FULL), segments of
# 1 / 2 / 4 / 8 MCU rows and the default; the fills overlapped with the sync passes (e2e); reduced-size decodes
O=gpurun_out/r3d; mkdir -p $O
ALT=$PWD/jpeg-decoder_amd/libjpgpu_alt.so
run() { # name, env..., -- bench args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-classes --no-e2e --no-k4096 --min-seconds 0 "$@" > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:22s} path {d['config']['kernel_path']:10s} ms {d['roofline']['kernel_ms_per_launch']:.4f} frac {d['roofline']['frac']:.4f} verified {d['verified_vs_oracle']}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2; do
  run row_default_$rep X=1 --
  run full_default_$rep JPGPU_LIBRARY=$ALT --
  for seg in 1 2 4 8; do
    run row_seg${seg}_$rep JPGPU_S420_SEG=$seg --
    run full_seg${seg}_$rep JPGPU_LIBRARY=$ALT JPGPU_S420_SEG=$seg --
  done
done
run 440_row X=1 -- --workload 1080p-440
run 440_full JPGPU_LIBRARY=$ALT -- --workload 1080p-440
run 440_row_seg4 JPGPU_S420_SEG=4 -- --workload 1080p-440
for wl in 1080p-420-scale4 1080p-420-scale2 1080p-420-scale1 1080p-444-scale4; do run $wl X=1 -- --workload $wl; done
timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 50 > $O/bench_e2e.json 2> $O/bench_e2e.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3d/bench_e2e.json").read().strip().splitlines()[-1])
for k in ("256", "4096"):
    e = d["e2e"][k]
    print(k, e["total_ms"], e["images_per_s"], e.get("kernel_ms"), e.get("kernels_only_images_per_s"), e["verified_vs_oracle"])
PY
timeout 900 python -m pytest tests -m gpu -q -x -k "420 or 440 or strip or pipeline or scale or compute_image or reftest" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 5 $O/pytest.log
