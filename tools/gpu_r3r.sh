#!/bin/bash
# round 3, call R: hardware queues of the HIP runtime (GPU_MAX_HW_QUEUES; default 4) against the pipeline's 8 compute + copy streams
O=gpurun_out/r3r; mkdir -p $O
run() { local name=$1; shift; env JPGPU_BATCH_KERNEL_TIMES=1 "$@" timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 --e2e-images 256,1024,4096 > $O/$name.json 2> $O/$name.err
python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "K ms", d["ms_per_step"], " ".join("E%s %.2f ms %.0f img/s" % (k, d["e2e"][k]["total_ms"], d["e2e"][k]["images_per_s"]) for k in ("256", "1024", "4096")), d["e2e"]["tower_progressive_256"]["images_per_s"])
PY
}
for q in 4 6 8 12 16 24; do run q$q GPU_MAX_HW_QUEUES=$q; done
run q8_sub512 GPU_MAX_HW_QUEUES=8 JPGPU_PIPE_DEV_SUB=512
run q16_sub128 GPU_MAX_HW_QUEUES=16 JPGPU_PIPE_DEV_SUB=128
run q8_t8 GPU_MAX_HW_QUEUES=8 JPGPU_SYNC_TAIL=8
