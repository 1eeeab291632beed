#!/bin/bash
# round 3, call U: does the number of hardware queues move the kernel-only figure K?  (interleaved, 3 repeats)
O=gpurun_out/r3u; mkdir -p $O
for rep in 1 2 3; do
for q in 4 8 10 16; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --no-classes --no-k4096 --no-e2e --steps 300 --warmup 50 --min-seconds 0 > $O/q${q}_$rep.json 2>>$O/err.txt
  python -c "
import json
d=json.loads(open('$O/q${q}_$rep.json').read().strip().splitlines()[-1])
print('q$q rep$rep', d['ms_per_step'], d['roofline']['frac'])"
done
done
