# same-box A/B of two builds of the library: libjpgpu.so vs libjpgpu_alt.so (JPGPU_LIBRARY), interleaved twice
#   usage: bash tools/ab_lib.sh <outdir> [bench args...]
O=gpurun_out/$1; shift
mkdir -p $O
B="python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-classes $@"
for rep in 1 2; do
  $B > $O/main_$rep.json 2>>$O/err.txt
  JPGPU_LIBRARY=$PWD/jpeg-decoder_amd/libjpgpu_alt.so $B > $O/alt_$rep.json 2>>$O/err.txt
  $B > $O/main_strip_$rep.json 2>>$O/err.txt
  JPGPU_LIBRARY=$PWD/jpeg-decoder_amd/libjpgpu_alt.so $B > $O/alt_strip_$rep.json 2>>$O/err.txt
done
for f in $O/*.json; do python -c "
import json,sys
l=json.load(open('$f'))
print('$f'.split('/')[-1], l['config']['kernel_path'], l['ms_per_step'], l['roofline']['frac'], l['verified_vs_oracle'])
"; done
