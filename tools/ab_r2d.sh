mkdir -p gpurun_out/r2d
B="python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-classes"
$B > gpurun_out/r2d/twopass.json 2>gpurun_out/r2d/err.txt
JPGPU_420_STRIP=1 JPGPU_S420_SEG=17 $B > gpurun_out/r2d/strip256_seg17.json 2>>gpurun_out/r2d/err.txt
for seg in 17 9 34; do JPGPU_420_STRIP=1 JPGPU_S420_SEG=$seg JPGPU_S420_TX=20 $B > gpurun_out/r2d/strip128_seg$seg.json 2>>gpurun_out/r2d/err.txt; done
JPGPU_F420_TX=40 $B > gpurun_out/r2d/twopass_tx40.json 2>>gpurun_out/r2d/err.txt
for f in gpurun_out/r2d/*.json; do echo $f; python -c "
import json,sys
l=json.load(open('$f'))
print(l['config']['kernel_path'], l['ms_per_step'], l['roofline']['frac'], l['verified_vs_oracle'])
"; done
tail -3 gpurun_out/r2d/err.txt
