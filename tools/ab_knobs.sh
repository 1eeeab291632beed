# same-box sweep of environment knobs: bash tools/ab_knobs.sh <outdir> <workload> "<K=V[,K2=V2]>" "<...>" ...
O=gpurun_out/$1; WL=$2; shift; shift
mkdir -p $O
B="python bench.py --steps ${STEPS:-300} --warmup 50 --no-cpu-baseline --no-classes --workload $WL"
$B > $O/base.json 2>>$O/err.txt
i=0
for knob in "$@"; do
  i=$((i+1))
  env $(echo $knob | tr ',' ' ') $B > $O/k${i}.json 2>>$O/err.txt
  echo "k$i = $knob" >> $O/knobs.txt
done
$B > $O/base2.json 2>>$O/err.txt
cat $O/knobs.txt
for f in $O/*.json; do python -c "
import json,sys
l=json.load(open('$f'))
print('$f'.split('/')[-1], l['config']['kernel_path'], l['ms_per_step'], l['roofline']['frac'], l['verified_vs_oracle'])
"; done
