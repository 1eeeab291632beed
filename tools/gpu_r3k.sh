#!/bin/bash
# round 3, call K: images per device-entropy sub-batch (JPGPU_PIPE_DEV_SUB) at 256 / 1024 / 4096 files per call
O=gpurun_out/r3k; mkdir -p $O
run() { local name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 --e2e-images 256,1024,4096 > $O/$name.json 2> $O/$name.err
python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("256", "1024", "4096"):
    e = d["e2e"][k]
    print(sys.argv[2], k, "total_ms", e["total_ms"], "img/s", e["images_per_s"], e.get("kernel_ms"), e["verified_vs_oracle"])
PY
}
run sub256 X=1
run sub512 JPGPU_PIPE_DEV_SUB=512
run sub1024 JPGPU_PIPE_DEV_SUB=1024
timeout 900 python -m pytest tests -m gpu -q -x -k "pipeline or decoder or concurrency" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 3 $O/pytest.log
