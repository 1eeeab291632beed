#!/bin/bash
# A/B pass: parity tests (filtered), then bench lines for "name ENV=..." rows in $AB_RUNS (WL=workload per row via env)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "${AB_TESTS:-batch or smoke or reftest or fused}" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -n 3 gpurun_out/pytest_gpu.log
run() { # name, env...
  name=$1; shift
  env "$@" bash -c 'timeout 600 python bench.py --steps ${STEPS:-500} --warmup ${WARM:-50} --workload ${WL:-1080p-420} --no-cpu-baseline' > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python -c "
import json
d=json.load(open('gpurun_out/ab_$name.json'))
print('$name', d['config']['kernel_path'], d['value'],'MP/s', d['roofline']['kernel_ms_per_launch'],'ms', d['roofline']['frac'], d['verified_vs_oracle'])
" 2>&1 | tail -1
}
while read -r name envs; do [ -n "$name" ] && run $name $envs; done <<< "$AB_RUNS"
