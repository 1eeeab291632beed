#!/bin/bash
# round 3, call H: sliced staging + upload of the device-entropy route; the generic colour kernel spanning rows
O=gpurun_out/r3h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 4 $O/pytest.log
for wl in 1080p-420-scale4 1080p-420-scale2 1080p-420-scale1 1080p-444-scale4; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-classes --min-seconds 0 >> $O/scaled.jsonl 2>> $O/scaled.err
done
timeout 300 python bench.py --generic --no-cpu-baseline --no-classes --min-seconds 0 --no-e2e --no-k4096 --steps 100 >> $O/scaled.jsonl 2>> $O/scaled.err
python - <<'PY'
import json
for l in open("gpurun_out/r3h/scaled.jsonl"):
    d = json.loads(l); print(f"{d['config']['name']:18s} {d['config']['kernel_path']:10s} ms {d['roofline']['kernel_ms_per_launch']:.4f} frac {d['roofline']['frac']:.4f} {d['verified_vs_oracle']}")
PY
for rep in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 > $O/e2e_$rep.json 2> $O/e2e_$rep.err
python - "$O/e2e_$rep.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("256", "4096", "tower_progressive_256"):
    e = d["e2e"][k]
    print(k, "total_ms", e["total_ms"], "img/s", e["images_per_s"], e.get("wall_ms"), e.get("kernel_ms"), e.get("kernels_only_images_per_s"), e["verified_vs_oracle"])
PY
done
