#!/bin/bash
# one-image device-entropy latency under chunk-size knobs
for k in "" "JPGPU_SYNC_BLOCKS=24" "JPGPU_SYNC_BLOCKS=12,JPGPU_SYNC_MIN_SHIFT=9,JPGPU_SYNC_LAUNCHES=16" "JPGPU_SYNC_BLOCKS=6,JPGPU_SYNC_MIN_SHIFT=8,JPGPU_SYNC_LAUNCHES=24" "JPGPU_SYNC_BLOCKS=3,JPGPU_SYNC_MIN_SHIFT=7,JPGPU_SYNC_LAUNCHES=32" "JPGPU_SYNC_BLOCKS=6,JPGPU_SYNC_MIN_SHIFT=8,JPGPU_SYNC_LAUNCHES=24,JPGPU_SYNC_ITERS=1"; do
  echo "== $k"
  env $(echo $k | tr ',' ' ') python tools/pipe1_timings.py 2>&1 | grep "device_entropy=True"
done
