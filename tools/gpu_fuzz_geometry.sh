#!/bin/bash
# bash tools/gpu_fuzz_geometry.sh <seeds...>   (each seed: 200 rounds with the default kernels, then 60 with the two-pass 4:2:0 kernels)
O=gpurun_out/fuzz_geometry; mkdir -p $O
for seed in "$@"; do
  timeout 900 python tools/fuzz_gpu_geometry.py $seed 200 2>&1 | tail -4 | tee -a $O/log.txt
  JPGPU_420_STRIP=0 timeout 900 python tools/fuzz_gpu_geometry.py $((seed + 1000)) 60 2>&1 | tail -4 | tee -a $O/log.txt
done
