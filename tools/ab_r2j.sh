mkdir -p gpurun_out/r2j
B="python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-classes"
run() { name=$1; shift; env "$@" $B > gpurun_out/r2j/$name.json 2>>gpurun_out/r2j/err.txt; }
run twopass A=1
run strip_seg17 JPGPU_420_STRIP=1 JPGPU_S420_SEG=17
run strip_seg9 JPGPU_420_STRIP=1 JPGPU_S420_SEG=9
run strip_seg23 JPGPU_420_STRIP=1 JPGPU_S420_SEG=23
run strip_seg34 JPGPU_420_STRIP=1 JPGPU_S420_SEG=34
run strip_seg12 JPGPU_420_STRIP=1 JPGPU_S420_SEG=12
run strip_seg17_3wg JPGPU_420_STRIP=1 JPGPU_S420_SEG=17 JPGPU_LDS_PAD=8000
run strip_seg17_2wg JPGPU_420_STRIP=1 JPGPU_S420_SEG=17 JPGPU_LDS_PAD=30000
run twopass_3wg JPGPU_LDS_PAD=8000
run strip_seg17_b JPGPU_420_STRIP=1 JPGPU_S420_SEG=17
for f in gpurun_out/r2j/*.json; do python -c "
import json,sys
l=json.load(open('$f'))
print('$f'.split('/')[-1], l['config']['kernel_path'], l['ms_per_step'], l['roofline']['frac'], l['verified_vs_oracle'])
"; done
