#!/bin/bash
# round 3, call S: compute streams in use (JPGPU_PIPE_STREAMS) with the runtime's default 4 hardware queues and with 16
O=gpurun_out/r3s; mkdir -p $O
run() { local name=$1; shift; env JPGPU_BATCH_KERNEL_TIMES=1 "$@" timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 --e2e-images 256,1024,4096 > $O/$name.json 2> $O/$name.err
python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "K ms", d["ms_per_step"], " ".join("E%s %.2f ms %.0f img/s" % (k, d["e2e"][k]["total_ms"], d["e2e"][k]["images_per_s"]) for k in ("256", "1024", "4096")), d["e2e"]["tower_progressive_256"]["images_per_s"])
PY
}
for s in 1 2 3 4 8; do run q4_s$s JPGPU_PIPE_STREAMS=$s; done
for s in 2 4 8; do run q16_s$s GPU_MAX_HW_QUEUES=16 JPGPU_PIPE_STREAMS=$s; done
run q16_s8_sub128 GPU_MAX_HW_QUEUES=16 JPGPU_PIPE_DEV_SUB=128
run q16_s8_sub192 GPU_MAX_HW_QUEUES=16 JPGPU_PIPE_DEV_SUB=192
