#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "four or worker or classes_decided or compute_image or reftest or golden or anchor" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 6 $O/pytest.log
for wl in 1080p-cmyk-2211 1080p-ycck-2212; do
  for tx in 16 12 8; do
  JPGPU_TX=$tx timeout 300 python bench.py --workload $wl --steps 200 --warmup 30 --no-cpu-baseline --no-classes --min-seconds 0 > $O/$wl-tx$tx.json 2> $O/$wl-tx$tx.err
  python - "$O/$wl-tx$tx.json" "$wl tx$tx" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:26s} path {d['config']['kernel_path']:16s} ms {d['roofline']['kernel_ms_per_launch']:.4f} frac {d['roofline']['frac']:.4f} verified {d['verified_vs_oracle']}")
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1][:-4] + "err").read()[-600:])
PY
  done
done
