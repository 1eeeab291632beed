#!/bin/bash
# round 3, call Q: 4096 files per call with emission: images per device sub-batch, hardware queues
O=gpurun_out/r3q; mkdir -p $O
run() { local name=$1; shift; env JPGPU_BATCH_KERNEL_TIMES=1 "$@" timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 --e2e-images 1024,4096 > $O/$name.json 2> $O/$name.err
python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("1024", "4096"):
    e = d["e2e"][k]
    print(sys.argv[2], k, "total_ms", e["total_ms"], "img/s", e["images_per_s"], e.get("kernel_ms"), e["verified_vs_oracle"])
PY
}
run sub256 X=1
run sub512 JPGPU_PIPE_DEV_SUB=512
run sub1024 JPGPU_PIPE_DEV_SUB=1024
run sub2048 JPGPU_PIPE_DEV_SUB=2048
run sub256_q8 GPU_MAX_HW_QUEUES=8
run sub256_t8 JPGPU_SYNC_TAIL=8
run sub1024_t8 JPGPU_PIPE_DEV_SUB=1024 JPGPU_SYNC_TAIL=8
