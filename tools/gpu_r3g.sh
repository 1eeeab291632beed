#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 5 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-k4096 --min-seconds 0 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3g/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_launch"], {k: (v.get("kernel_ms_per_launch"), v.get("frac")) for k, v in d["roofline_by_class"].items() if isinstance(v, dict)})
PY
for wl in 1080p-440 1080p-cmyk-2211 1080p-ycck-2212; do
timeout 300 python bench.py --workload $wl --no-cpu-baseline --min-seconds 0 --steps 200 > $O/$wl.json 2> $O/$wl.err
python - "$O/$wl.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["name"], d["roofline"]["frac"], d["roofline"]["kernel_ms_per_launch"], {k: (v.get("kernel_ms_per_launch")) for k, v in d["roofline_by_class"].items() if isinstance(v, dict)})
PY
done
