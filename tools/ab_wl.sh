# same-box A/B of libjpgpu.so vs libjpgpu_alt.so over a list of workloads: bash tools/ab_wl.sh <outdir> "<wl1 wl2 ...>" [reps]
O=gpurun_out/$1; WLS="$2"; REPS=${3:-2}
mkdir -p $O
for wl in $WLS; do
  B="python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-classes --workload $wl"
  for rep in $(seq 1 $REPS); do
    $B > $O/main_${wl}_$rep.json 2>>$O/err.txt
    JPGPU_LIBRARY=$PWD/jpeg-decoder_amd/libjpgpu_alt.so $B > $O/alt_${wl}_$rep.json 2>>$O/err.txt
  done
done
for f in $O/*.json; do python -c "
import json,sys
l=json.load(open('$f'))
print('$f'.split('/')[-1], l['config']['kernel_path'], l['ms_per_step'], l['roofline']['frac'], l['verified_vs_oracle'])
"; done
