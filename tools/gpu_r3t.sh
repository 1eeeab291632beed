#!/bin/bash
# round 3, call T: images per device sub-batch x most sub-batches per call, 16 hardware queues
O=gpurun_out/r3t; mkdir -p $O
run() { local name=$1; shift; env JPGPU_BATCH_KERNEL_TIMES=1 GPU_MAX_HW_QUEUES=16 "$@" timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 --e2e-images 256,1024,4096 > $O/$name.json 2> $O/$name.err
python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "K ms", d["ms_per_step"], " ".join("E%s %.2f ms %.0f img/s" % (k, d["e2e"][k]["total_ms"], d["e2e"][k]["images_per_s"]) for k in ("256", "1024", "4096")), d["e2e"]["tower_progressive_256"]["images_per_s"])
PY
}
run sub256_cap16 X=1
run sub128_cap16 JPGPU_PIPE_DEV_SUB=128
run sub128_cap32 JPGPU_PIPE_DEV_SUB=128 JPGPU_PIPE_MAX_DEV_SUBS=32
run sub64_cap32 JPGPU_PIPE_DEV_SUB=64 JPGPU_PIPE_MAX_DEV_SUBS=32
run sub64_cap16 JPGPU_PIPE_DEV_SUB=64
run sub96_cap32 JPGPU_PIPE_DEV_SUB=96 JPGPU_PIPE_MAX_DEV_SUBS=32
run sub128_cap16_b JPGPU_PIPE_DEV_SUB=128
run sub256_cap16_b X=1
