O=gpurun_out/seam1; mkdir -p $O
B="python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-classes"
A=$PWD/jpeg-decoder_amd/libjpgpu_alt.so
for rep in 1 2; do
$B > $O/main_$rep.json 2>>$O/err.txt
JPGPU_LIBRARY=$A $B > $O/stub_seg23_$rep.json 2>>$O/err.txt
for sg in 2 3 4 6 9; do JPGPU_LIBRARY=$A JPGPU_S420_SEG=$sg $B > $O/stub_seg${sg}_$rep.json 2>>$O/err.txt; done
done
for f in $O/*.json; do python -c "
import json,sys
l=json.load(open('$f'))
print('$f'.split('/')[-1], l['config']['kernel_path'], l['ms_per_step'], l['roofline']['frac'], l['verified_vs_oracle'])
"; done
