#!/bin/bash
# round 3, call C: full GPU tests (incl. 1,024 decoding threads), the default bench line, and the layouts still on the generic path
O=gpurun_out/r3c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 8 $O/pytest.log
for wl in 1080p-420-scale4 1080p-420-scale2 1080p-420-scale1 1080p-444-scale4 1080p-cmyk-2211 1080p-ycck-2212; do
  timeout 300 python bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline --no-classes --min-seconds 0 > $O/$wl.json 2> $O/$wl.err
  python - "$O/$wl.json" "$wl" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:20s} path {d['config']['kernel_path']:10s} ms {d['roofline']['kernel_ms_per_launch']:.4f} frac {d['roofline']['frac']:.4f} verified {d['verified_vs_oracle']}")
except Exception as e:
    print(sys.argv[2], "failed", e, open(sys.argv[1][:-4] + "err").read()[-400:])
PY
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3c/bench_driver.json").read().strip().splitlines()[-1])
print(json.dumps({k: d.get(k) for k in ("value", "ms_per_step", "e2e", "cpu_baseline_e2e")})[:3500])
PY
