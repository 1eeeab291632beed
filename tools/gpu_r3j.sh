#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
run() { local name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 > $O/$name.json 2> $O/$name.err
python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("256", "4096"):
    e = d["e2e"][k]
    print(sys.argv[2], k, "total_ms", e["total_ms"], "img/s", e["images_per_s"], e.get("kernel_ms"), e["verified_vs_oracle"])
PY
}
run base X=1
run assemble JPGPU_SYNC_WRITE_ASSEMBLE=1
run wlds0 JPGPU_SYNC_WRITE_LDS=16384
run launches6 JPGPU_SYNC_LAUNCHES=6
