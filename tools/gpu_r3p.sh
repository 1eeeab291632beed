#!/bin/bash
# round 3, call P: first sync pass tail-only for every lane (the first one too); where the host time of a 4096-file call goes
O=gpurun_out/r3p; mkdir -p $O
run() { local name=$1; shift; env JPGPU_BATCH_KERNEL_TIMES=1 "$@" timeout 600 python bench.py --no-cpu-baseline --no-classes --no-k4096 --steps 30 --min-seconds 0 --e2e-images 256 > $O/$name.json 2> $O/$name.err
python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k in ("256",):
    e = d["e2e"][k]
    print(sys.argv[2], k, "total_ms", e["total_ms"], "img/s", e["images_per_s"], e.get("kernel_ms"), e["verified_vs_oracle"])
PY
}
for tail in 2 3 4 5 8; do
  for iters in 1 2; do
    run t${tail}_i${iters} JPGPU_SYNC_TAIL=$tail JPGPU_SYNC_ITERS=$iters JPGPU_SYNC_LAUNCHES=12
  done
done
JPGPU_SYNC_TAIL=3 JPGPU_SYNC_ITERS=1 bash tools/gpu_trace_seq.sh 18 python $GRAFT_REPO_ROOT/tools/e2e_bench.py --images 256 --device-entropy --no-download --rounds 3
JPGPU_PIPE_TRACE=1 timeout 300 python tools/e2e_bench.py --images 4096 --device-entropy --no-download --rounds 3 > $O/trace4096.txt 2>&1
grep -c "pipeline trace" $O/trace4096.txt; tail -n 45 $O/trace4096.txt
