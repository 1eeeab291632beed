#!/bin/bash
# quick GPU pass: batch parity tests + benches + kernel trace + PMC passes
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "batch or smoke or reftest" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
for wl in 1080p-420 1080p-444 1080p-gray 2160p-420; do
  timeout 600 python bench.py --steps 20 --warmup 3 --workload $wl --no-cpu-baseline > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err
done
for st in 2 3 4; do JPGPU_STREAMS=$st timeout 600 python bench.py --steps 20 --warmup 3 --workload 1080p-420 --no-cpu-baseline > gpurun_out/bench_1080p-420-st$st.json 2> gpurun_out/bench_st.err; done
JPGPU_STREAMS=2 JPGPU_CHUNK=32 timeout 600 python bench.py --steps 20 --warmup 3 --workload 1080p-420 --no-cpu-baseline > gpurun_out/bench_1080p-420-st2c32.json 2> gpurun_out/bench_st.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_quick -o q -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_quick.log 2>&1
if [ -n "$PMC" ]; then
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmc1 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc3 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc4 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc4.log 2>&1
fi
cd $R
python tools/prof_summary.py gpurun_out/prof_quick gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 gpurun_out/pmc4 2>&1 | tail -40
tail -n 3 gpurun_out/pytest_gpu.log
for wl in 1080p-420 1080p-420-st2 1080p-420-st3 1080p-420-st4 1080p-420-st2c32 1080p-444 1080p-gray 2160p-420; do python -c "
import json,sys
d=json.load(open('gpurun_out/bench_$wl.json'))
print('$wl', d['config']['kernel_path'], d['value'],'MP/s', d['roofline']['kernel_ms_per_launch'],'ms', d['roofline']['achieved'],'GB/s', d['roofline']['frac'], d['verified_vs_oracle'])
" 2>&1 | tail -1; done
