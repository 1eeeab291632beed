#!/bin/bash
# how often does a device entropy launch stall on the host (JPGPU_PIPE_TRACE: launches over 3 ms after the cold call)?  tools/gpu_stalls.sh <tag> [ENV=VAL ...]
tag=$1; shift
env JPGPU_PIPE_TRACE=1 "$@" timeout 300 python tools/e2e_bench.py --images 4096 --device-entropy --no-download --rounds 16 > gpurun_out/stalls_$tag.txt 2>&1
echo "$tag: warm stalls $(grep "slow device" gpurun_out/stalls_$tag.txt | grep -vc "buffers [0-9][0-9]\.\|buffers [2-9]\.") of $(grep -c "entropy launch of" gpurun_out/stalls_$tag.txt) launches; $(tail -1 gpurun_out/stalls_$tag.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('best', d['ms']['total_ms'], 'ms; sustained', d['sustained_images_per_s_pixels_left_in_hbm'], 'img/s')")"
grep "slow device" gpurun_out/stalls_$tag.txt | grep -v "buffers [0-9][0-9]\.\|buffers [2-9]\." | cut -c1-200 | head -4
