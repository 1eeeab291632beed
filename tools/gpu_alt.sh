#!/bin/bash
# E figures + standalone kernel phases for libjpgpu.so and every jpeg-decoder_amd/libjpgpu_alt*.so: tools/gpu_alt.sh <outdir>
O=$1
NAME=main bash tools/gpu_e.sh $O
for lib in jpeg-decoder_amd/libjpgpu_alt*.so; do
  NAME=$(basename $lib .so) bash tools/gpu_e.sh $O JPGPU_LIBRARY=$PWD/$lib
done
NAME=main2 bash tools/gpu_e.sh $O
