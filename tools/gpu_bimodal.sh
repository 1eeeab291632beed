#!/bin/bash
# the 256-file call is 6.7 ms in one process and 9.0-9.7 ms in the next: host side or device side?  (JPGPU_PIPE_TRACE of several processes)
for k in 1 2 3 4 5 6; do
  JPGPU_PIPE_TRACE=1 JPGPU_BATCH_KERNEL_TIMES=1 python tools/e256_calls.py > gpurun_out/bimodal_$k.txt 2>&1
  tail -1 gpurun_out/bimodal_$k.txt | cut -c1-120
  grep "entropy launch of" gpurun_out/bimodal_$k.txt | tail -8 | awk '{s+=$13} END {printf "   host ms per launch %.2f;", s/NR}'
  grep "on the device" gpurun_out/bimodal_$k.txt | tail -8 | awk '{print $11, $14, $16, $19, $23, $25}' | tr -d ',' | awk '{sy+=$3-$2; ex+=$4-$3; px+=$6-$5; last=$6} END {printf " device: sync %.2f expand %.2f pixel %.2f ms per sub-batch, last pixel end +%.2f\n", sy/NR, ex/NR, px/NR, last}'
done
