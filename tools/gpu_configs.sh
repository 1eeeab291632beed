#!/bin/bash
# SURVEY §8d configurations that are not the default bench line
mkdir -p gpurun_out
timeout 600 python bench.py --workload 1080p-444+gray --no-cpu-baseline > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; cut -c1-900 gpurun_out/bench_c5.json; tail -2 gpurun_out/bench_c5.err
timeout 600 python bench.py --workload 2160p-420 --batch 512 --no-cpu-baseline > gpurun_out/bench_c3_1gpu.json 2> gpurun_out/bench_c3.err; cut -c1-200,560-900 gpurun_out/bench_c3_1gpu.json; tail -2 gpurun_out/bench_c3.err
timeout 600 python tools/e2e_bench.py --file tests/golden/benches/tower_progressive.jpg --images 256 > gpurun_out/e2e_c4.json 2>&1; tail -1 gpurun_out/e2e_c4.json
timeout 600 python tools/e2e_bench.py --file tests/golden/benches/tower.jpg --images 256 > gpurun_out/e2e_c1x256.json 2>&1; tail -1 gpurun_out/e2e_c1x256.json
timeout 600 python tools/e2e_bench.py --file tests/golden/benches/tower.jpg --images 1 --threads 1 > gpurun_out/e2e_c1.json 2>&1; tail -1 gpurun_out/e2e_c1.json
timeout 600 python tools/e2e_bench.py --file tests/golden/benches/tower_progressive.jpg --images 256 --progressive-deltas > gpurun_out/e2e_c4_deltas.json 2>&1; tail -1 gpurun_out/e2e_c4_deltas.json
timeout 600 python tools/e2e_bench.py --file tests/golden/benches/tower.jpg --images 256 --device-entropy > gpurun_out/e2e_c1x256_dev.json 2>&1; tail -1 gpurun_out/e2e_c1x256_dev.json
