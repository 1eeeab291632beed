#!/bin/bash
O=gpurun_out/fuzz_files; mkdir -p $O
for seed in "$@"; do timeout 900 python tools/fuzz_gpu_files.py $seed 96 2>&1 | tail -6 | tee -a $O/log.txt; done
