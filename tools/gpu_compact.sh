#!/bin/bash
# huff_sync_pass_kernel<4> (four table slots in LDS: 24 kB, six workgroups per CU) against <8> (40 kB, four): tools/gpu_compact.sh <outdir>
O=$1
NAME=compact bash tools/gpu_e.sh $O
NAME=full bash tools/gpu_e.sh $O JPGPU_SYNC_COMPACT_TABLES=0
NAME=compact2 bash tools/gpu_e.sh $O
NAME=full2 bash tools/gpu_e.sh $O JPGPU_SYNC_COMPACT_TABLES=0
