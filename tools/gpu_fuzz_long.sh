#!/bin/bash
# A longer differential fuzzing campaign of the device entropy route (tools/fuzz_gpu.py): seeds x 200 damaged variants of each of the 12 bases.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/fuzzlong
out=gpurun_out/fuzzlong/fuzz.txt
: > $out
for seed in "$@"; do
  echo "== seed $seed" >> $out
  s=$(date +%s); timeout 600 python tools/fuzz_gpu.py $seed 200 >> $out 2>&1; echo "$(( $(date +%s) - s )) s" >> $out
done
tail -40 $out
