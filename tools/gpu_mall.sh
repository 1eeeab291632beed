#!/bin/bash
# does a small sub-batch's pixel kernel find the coefficients the expansion just wrote in the infinity cache (256 MB)?  kernel phases of ONE sub-batch of n images
for n in 16 32 64 128 256; do
  JPGPU_BATCH_KERNEL_TIMES=1 JPGPU_PIPE_DEV_SUB=$n JPGPU_PIPE_MAX_DEV_SUBS=1 timeout 200 python tools/e2e_bench.py --images $n --device-entropy --no-download --rounds 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); m = d['ms']; n = d['images']
print('images %4d  sync %.3f  expand %.3f  pixel %.3f ms   per 256 images: expand %.3f pixel %.3f' % (n, m['dev_sync_ms'], m['dev_write_ms'], m['dev_pixel_ms'], m['dev_write_ms'] * 256 / n, m['dev_pixel_ms'] * 256 / n))"
done
