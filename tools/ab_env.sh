#!/bin/bash
# same-box A/B of an environment knob: bash tools/ab_env.sh <outdir> "<VAR=value ...>" [bench args...]
O=gpurun_out/$1; KNOB="$2"; shift; shift
mkdir -p $O
B="python bench.py --steps 300 --warmup 50 --no-cpu-baseline $@"
for rep in 1 2; do
  $B > $O/base_$rep.json 2>>$O/err.txt
  env $KNOB $B > $O/knob_$rep.json 2>>$O/err.txt
done
for f in $O/*.json; do python -c "
import json
l=json.load(open('$f'))
c=l.get('roofline_by_class',{})
print('$f'.split('/')[-1], l['config']['kernel_path'], l['ms_per_step'], l['roofline']['frac'], l['verified_vs_oracle'], {k:v['kernel_ms_per_launch'] for k,v in c.items() if isinstance(v,dict) and 'kernel_ms_per_launch' in v})
"; done; tail -2 $O/err.txt
