#!/bin/bash
# round 3, call V: defaults after call T/U (16 hardware queues set by bench.py itself, 128 images per device sub-batch, up to 32 of them): full GPU tests + the driver's command
O=gpurun_out/r3v; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 4 $O/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r3v/bench_driver.json').read().strip().splitlines()[-1])
print("K", d["value"], d["ms_per_step"], d["roofline"]["frac"], "k4096", d.get("k_4096", {}).get("ms_per_step"))
for k, e in d["e2e"].items():
    if isinstance(e, dict) and "images_per_s" in e: print("E", k, e["total_ms"], e["images_per_s"], e.get("kernel_ms"), e.get("kernels_only_images_per_s"), e["verified_vs_oracle"])
print("cpu e2e", d.get("cpu_baseline_e2e"))
PY
