#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "pipeline or decoder or concurrency" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 3 $O/pytest.log
for rep in 1 2 3; do
timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 > $O/e2e_$rep.json 2> $O/e2e_$rep.err
python - "$O/e2e_$rep.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("256", "4096"):
    e = d["e2e"][k]
    print(k, "total_ms", e["total_ms"], "img/s", e["images_per_s"], e.get("wall_ms"), e.get("kernel_ms"), e.get("kernels_only_images_per_s"), e["verified_vs_oracle"])
PY
done
