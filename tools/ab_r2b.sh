# A/B on one box: two-pass 4:2:0 vs the single-launch strip walk at several segment lengths; other kinds after the IDCT changes
mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "strip or full_size or same_geometry" 2>&1 | tail -5 > gpurun_out/r2b/tests.txt
B="python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-classes"
$B > gpurun_out/r2b/twopass.json 2>gpurun_out/r2b/err.txt
for seg in 0 9 17 34 68; do JPGPU_420_STRIP=1 JPGPU_S420_SEG=$seg $B > gpurun_out/r2b/strip_seg$seg.json 2>>gpurun_out/r2b/err.txt; done
JPGPU_420_STRIP=1 JPGPU_S420_SEG=17 JPGPU_S420_TX=30 $B > gpurun_out/r2b/strip_seg17_tx30.json 2>>gpurun_out/r2b/err.txt
$B --workload 1080p-422 > gpurun_out/r2b/422.json 2>>gpurun_out/r2b/err.txt
$B --workload 1080p-444 > gpurun_out/r2b/444.json 2>>gpurun_out/r2b/err.txt
$B --workload 1080p-gray > gpurun_out/r2b/gray.json 2>>gpurun_out/r2b/err.txt
cat gpurun_out/r2b/tests.txt
for f in gpurun_out/r2b/*.json; do echo $f; python -c "
import json,sys
l=json.load(open('$f'))
print(l['config']['kernel_path'], l['ms_per_step'], l['roofline']['frac'], l['verified_vs_oracle'])
"; done
tail -5 gpurun_out/r2b/err.txt
