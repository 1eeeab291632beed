#!/bin/bash
cat /sys/fs/cgroup/cpu.max 2>/dev/null
for th in 0 16 8; do
timeout 300 python tools/e2e_bench.py --device-entropy --images 1024 --no-download --threads $th 2>&1 | tail -1 | cut -c230-520
done
timeout 300 python tools/e2e_bench.py --images 256 --no-download 2>&1 | tail -1 | cut -c230-520
JPGPU_PIPE_TRACE=1 timeout 300 python tools/e2e_bench.py --device-entropy --images 1024 --no-download --rounds 2 2>&1 | grep "pipeline trace" | tail -14
