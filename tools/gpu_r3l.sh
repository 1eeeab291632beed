#!/bin/bash
# round 3, call L: speculative emission (JPGPU_SYNC_EMIT=1, the default) against the write pass (=0): parity first, then E with phase times
O=gpurun_out/r3l; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -k "pipeline or decoder or concurrency or entropy or anchor" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 5 $O/pytest.log
run() { local name=$1; shift; env JPGPU_BATCH_KERNEL_TIMES=1 "$@" timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 --e2e-images 256,4096 > $O/$name.json 2> $O/$name.err
python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("256", "4096"):
    e = d["e2e"][k]
    print(sys.argv[2], k, "total_ms", e["total_ms"], "img/s", e["images_per_s"], e.get("kernel_ms"), e.get("kernel_phases_ms"), e["verified_vs_oracle"])
PY
}
run emit X=1
run write JPGPU_SYNC_EMIT=0
