#!/bin/bash
# E figures + the kernel phases of one sub-batch of 256 alone: tools/gpu_e.sh <outdir> [ENV=VAL ...]
O=gpurun_out/$1; shift; mkdir -p $O
name=${NAME:-run}
env "$@" timeout 900 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 --e2e-images 256,1024,4096 > $O/$name.json 2> $O/$name.err
python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e = d["e2e"]
print(sys.argv[2], "K ms", d["ms_per_step"], " ".join("E%s %.2f ms %.0f img/s" % (k, e[k]["total_ms"], e[k]["images_per_s"]) for k in ("256", "1024", "4096") if k in e),
      "| prog", e["tower_progressive_256"]["images_per_s"], "| alone", e.get("kernels_256_one_sub_batch", {}).get("kernel_ms"), all(e[k]["verified_vs_oracle"] for k in ("256", "1024", "4096") if k in e))
PY
