#!/bin/bash
# round 3, call E: GPU tests with the four-component row kernel; its bench lines; sub-batch size of the device-entropy route
O=gpurun_out/r3e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 12 $O/pytest.log
for wl in 1080p-cmyk-2211 1080p-ycck-2212 1080p-cmyk; do
  timeout 300 python bench.py --workload $wl --steps 200 --warmup 30 --no-cpu-baseline --min-seconds 0 > $O/$wl.json 2> $O/$wl.err
  python - "$O/$wl.json" "$wl" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    bc = {k: v.get("kernel_ms_per_launch") for k, v in d.get("roofline_by_class", {}).items() if isinstance(v, dict)}
    print(f"{sys.argv[2]:20s} path {d['config']['kernel_path']:16s} ms {d['roofline']['kernel_ms_per_launch']:.4f} frac {d['roofline']['frac']:.4f} verified {d['verified_vs_oracle']} {bc}")
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1][:-4] + "err").read()[-600:])
PY
done
for sub in 0 128 64; do
  JPGPU_PIPE_DEV_SUB=$sub timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 > $O/e2e_sub$sub.json 2> $O/e2e_sub$sub.err
  python - "$O/e2e_sub$sub.json" "$sub" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    for k in ("256", "4096"):
        e = d["e2e"][k]
        print("dev_sub", sys.argv[2], k, "total_ms", e["total_ms"], "img/s", e["images_per_s"], e.get("kernel_ms"), e["verified_vs_oracle"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
