#!/bin/bash
# round 3, call B: balanced strip walks (one share per resident workgroup) against round 2's segments, same box
O=gpurun_out/r3b; mkdir -p $O
run() { # name, env..., -- bench args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-classes --no-e2e --no-k4096 --min-seconds 0 "$@" > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:28s} ms {d['roofline']['kernel_ms_per_launch']:.4f} frac {d['roofline']['frac']:.4f} verified {d['verified_vs_oracle']}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2; do
run seg_$rep JPGPU_WALK_BALANCE=0 --
run bal1_$rep JPGPU_WALK_ROUNDS=1 --
run bal2_$rep JPGPU_WALK_ROUNDS=2 --
run bal3_$rep JPGPU_WALK_ROUNDS=3 --
done
run 2160_seg JPGPU_WALK_BALANCE=0 -- --workload 2160p-420
run 2160_bal1 JPGPU_WALK_ROUNDS=1 -- --workload 2160p-420
run 2160_bal2 JPGPU_WALK_ROUNDS=2 -- --workload 2160p-420
run 440_seg JPGPU_WALK_BALANCE=0 -- --workload 1080p-440
run 440_bal1 JPGPU_WALK_ROUNDS=1 -- --workload 1080p-440
run 440_bal2 JPGPU_WALK_ROUNDS=2 -- --workload 1080p-440
run b64_seg JPGPU_WALK_BALANCE=0 -- --batch 64
run b64_bal1 JPGPU_WALK_ROUNDS=1 -- --batch 64
run b1000_seg JPGPU_WALK_BALANCE=0 -- --batch 1000
run b1000_bal1 JPGPU_WALK_ROUNDS=1 -- --batch 1000
timeout 900 python -m pytest tests -m gpu -q -x -k "420 or 440 or strip or hostile or classes or compact or worker or mixed" > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -n 6 $O/pytest.log
