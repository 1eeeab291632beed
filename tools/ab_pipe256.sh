#!/bin/bash
# per-kernel GPU time of a 256 x 1080p device-entropy pipeline call: libjpgpu.so and every libjpgpu_alt*.so
for lib in "" jpeg-decoder_amd/libjpgpu_alt*.so; do
  echo "=== ${lib:-libjpgpu.so}"
  if [ -n "$lib" ]; then export JPGPU_LIBRARY=$PWD/$lib; else unset JPGPU_LIBRARY; fi
  bash tools/gpu_pipe256_trace.sh 2>&1 | grep -E "call ms|us  x" | cut -c1-110 | head -8
done
