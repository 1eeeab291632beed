#!/bin/bash
mkdir -p gpurun_out
for sw in "20 3" "100 10" "500 20" "2000 50" "500 300"; do set -- $sw
  timeout 600 python bench.py --steps $1 --warmup $2 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/steps_$1_$2.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/steps_$1_$2.json'))
print('steps $1 warmup $2:', d['value'],'MP/s', d['ms_per_step'], 'ms/step; kernel', d['roofline']['kernel_ms_per_launch'],'ms', d['roofline']['frac'])"
done
