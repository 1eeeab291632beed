#!/bin/bash
# Produces the artefacts summarised under profiles/: bench line, rocprofv3 kernel-trace stats and the
# PMC passes (each --pmc set in its own run, with --kernel-trace only).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
for wl in 1080p-444 1080p-422 1080p-gray 2160p-420; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err
done
timeout 600 python bench.py --steps 50 --warmup 10 --generic --no-cpu-baseline > gpurun_out/bench_generic.json 2> gpurun_out/bench_generic.err
cd /tmp
CMD="python $R/bench.py --no-cpu-baseline"            # same command as the bench line (500 steps after 50 warm-up)
PCMD="python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline"  # counter passes serialise dispatches: fewer launches
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o f -- $CMD > $R/gpurun_out/prof_final.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmc1 -o p -- $PCMD > $R/gpurun_out/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2 -o p -- $PCMD > $R/gpurun_out/pmc2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc3 -o p -- $PCMD > $R/gpurun_out/pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc4 -o p -- $PCMD > $R/gpurun_out/pmc4.log 2>&1
cd $R
cat gpurun_out/bench_default.json
python tools/prof_summary.py gpurun_out/prof_final gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 gpurun_out/pmc4 > gpurun_out/final_summary.json; tail -c 600 gpurun_out/final_summary.json
cd tools && python make_pmc_traffic.py 1080p-420:fused420 ../gpurun_out/pmc3 ../gpurun_out/pmc4 ../gpurun_out/pmc_traffic.json f420_ && cd ..
