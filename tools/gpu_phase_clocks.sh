#!/bin/bash
# diagnostic: per-phase wave time of the strip walk (build: make BUILD=build_alt OUT=../libjpgpu_alt.so CXXFLAGS="... -DJPGPU_PHASE_CLOCKS")
O=gpurun_out/phase_clocks; mkdir -p $O
JPGPU_LIBRARY=$PWD/jpeg-decoder_amd/libjpgpu_alt.so python tools/phase_clocks.py "$@" > $O/420.txt 2> $O/err.txt
cat $O/420.txt; tail -3 $O/err.txt
