#!/bin/bash
# The round's closing call on a GPU box, in the order VERDICT r5 #8 asks for — tests -> traffic (installed where the bench step of the
# same call reads it) -> kernel trace -> bench (the driver's command) -> force-dist -> the rest — into gpurun_out/<name>:
#   git rev-parse HEAD > tools/.head_commit; gpurun --timeout 4500 -- bash tools/closing_run.sh r6f
#   python tools/assemble_profiles.py gpurun_out/r6f        (here: composes profiles/round6 and its README from it)
N=${1:-closing}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$N; mkdir -p $O
export TRAFFIC_INSTALL=profiles/round6/pmc_traffic.json
bash tools/gpu.sh $N tests
bash tools/gpu.sh $N traffic:1080p-420:fused420:s420_kernel
bash tools/gpu.sh $N "trace:bench_K:python+bench.py+--no-e2e+--no-k4096+--no-cpu-baseline+--no-scale-anchor+--min-seconds+0"
bash tools/gpu.sh $N bench
bash tools/gpu.sh $N forcedist
bash tools/gpu.sh $N "pipe256:JPGPU_PIPE_ENTRY_PIXELS=1"
export JPGPU_PIPE_PROG_DEVICE_PERCENT=100
bash tools/gpu.sh $N "trace:prog256:python+tools/prog_calls.py+--images+256+--calls+4" "pmc:prog256:python+tools/prog_calls.py+--images+256+--calls+3" > $O/prog_prof.log 2>&1
unset JPGPU_PIPE_PROG_DEVICE_PERCENT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_scalar_chain.bin tools/ubench_scalar_chain.hip 2>/dev/null && /tmp/ubench_scalar_chain.bin > $O/ubench_scalar_chain.txt 2>&1
timeout 300 python tools/progw_asm_bench.py > $O/progw_asm_bench.txt 2>&1
( for n in 64 128 192 256 1024 4096; do for pc in 100 0; do [ $n = 4096 ] && [ $pc = 0 ] && continue; echo "== $n frames, device percent $pc"; timeout 300 python tools/prog_calls.py --images $n --calls 4 --percent $pc 2>&1 | tail -2; done; done
  echo "== 4096 distinct frames, device"; timeout 300 python tools/prog_calls.py --images 4096 --calls 4 --percent 100 --distinct 2>&1 | tail -2
  echo "== the dispatcher's own choice"; for n in 64 128 192 256 4096; do JPGPU_PIPE_TRACE=1 timeout 300 python tools/prog_calls.py --images $n --calls 2 --percent auto 2>&1 | grep "dispatcher:" | tail -1; done
  echo "== per scan (JPGPU_PROG_TIMES=1, 256 frames)"; JPGPU_PROG_TIMES=1 timeout 300 python tools/prog_calls.py --images 256 --calls 2 --percent 100 2>&1 | tail -15 ) > $O/progressive_calls.txt 2>&1
bash tools/gpu.sh $N fuzz:300
( for wl in 1080p-444 1080p-422 1080p-440 1080p-411 1080p-gray 1080p-cmyk 1080p-cmyk-2211 1080p-ycck-2212 1080p-420-scale4 1080p-420-scale2 1080p-420-scale1 1080p-444-scale4 1080p-444+gray 2160p-420; do timeout 300 python bench.py --workload $wl --steps 200 --warmup 30 --no-cpu-baseline --no-classes --min-seconds 0 2>/dev/null | tail -1; done ) > $O/other_workloads_bench.jsonl
cp profiles/round6/pmc_traffic.json $O/pmc_traffic_installed.json
