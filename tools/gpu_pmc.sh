#!/bin/bash
# PMC passes of the default bench command (env knobs pass through); summary on stdout
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
CMD="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS}"
rm -rf $R/gpurun_out/pmcq*
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmcq1 -o p -- $CMD > $R/gpurun_out/pmcq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $R/gpurun_out/pmcq2 -o p -- $CMD > $R/gpurun_out/pmcq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmcq3 -o p -- $CMD > $R/gpurun_out/pmcq3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmcq4 -o p -- $CMD > $R/gpurun_out/pmcq4.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/pmcq1 gpurun_out/pmcq2 gpurun_out/pmcq3 gpurun_out/pmcq4
