# same-box comparison of libjpgpu.so and every jpeg-decoder_amd/libjpgpu_alt*.so: bash tools/ab_libs.sh <outdir> "<workloads>"
O=gpurun_out/$1; WLS="$2"
mkdir -p $O
for wl in $WLS; do
  B="python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-classes --workload $wl"
  $B > $O/main_${wl}.json 2>>$O/err.txt
  for lib in jpeg-decoder_amd/libjpgpu_alt*.so; do
    n=$(basename $lib .so)
    JPGPU_LIBRARY=$PWD/$lib $B > $O/${n}_${wl}.json 2>>$O/err.txt
  done
  $B > $O/main2_${wl}.json 2>>$O/err.txt
done
for f in $O/*.json; do python -c "
import json,sys
l=json.load(open('$f'))
print('$f'.split('/')[-1], l['config']['kernel_path'], l['ms_per_step'], l['roofline']['frac'], l['verified_vs_oracle'])
"; done
