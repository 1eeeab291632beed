# same-box A/B of two builds over the other fused kinds
O=gpurun_out/$1; shift
mkdir -p $O
for wl in 1080p-444 1080p-422; do
  B="python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-classes --workload $wl"
  for rep in 1 2 3; do
    $B > $O/main_${wl}_$rep.json 2>>$O/err.txt
    JPGPU_LIBRARY=$PWD/jpeg-decoder_amd/libjpgpu_alt.so $B > $O/alt_${wl}_$rep.json 2>>$O/err.txt
  done
done
for f in $O/*.json; do python -c "
import json,sys
l=json.load(open('$f'))
print('$f'.split('/')[-1], l['config']['kernel_path'], l['ms_per_step'], l['roofline']['frac'], l['verified_vs_oracle'])
"; done
