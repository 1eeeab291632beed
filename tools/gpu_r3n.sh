#!/bin/bash
# round 3, call N: chunk size / iterations per launch with emission (the partial passes are latency chains: one lane = one chunk)
O=gpurun_out/r3n; mkdir -p $O
run() { local name=$1; shift; env JPGPU_BATCH_KERNEL_TIMES=1 "$@" timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 --e2e-images 256 > $O/$name.json 2> $O/$name.err
python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("256",):
    e = d["e2e"][k]
    print(sys.argv[2], k, "total_ms", e["total_ms"], "img/s", e["images_per_s"], e.get("kernel_ms"), e["verified_vs_oracle"])
PY
}
for blocks in 12 16 24 32 48; do
  for iters in 1 2 4; do
    run b${blocks}_i${iters} JPGPU_SYNC_BLOCKS=$blocks JPGPU_SYNC_ITERS=$iters JPGPU_SYNC_LAUNCHES=16
  done
done
