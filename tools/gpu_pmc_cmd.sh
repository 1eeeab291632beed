#!/bin/bash
# PMC pass (SQ counters) of an arbitrary command: tools/gpu_pmc_cmd.sh <cmd...>
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/pmcc
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD -d $R/gpurun_out/pmcc -o p -- "$@" > $R/gpurun_out/pmcc.log 2>&1
rm -rf $R/gpurun_out/pmcd
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAIT_ANY -d $R/gpurun_out/pmcd -o p -- "$@" > $R/gpurun_out/pmcd.log 2>&1
cd $R
python tools/prof_summary.py gpurun_out/pmcc gpurun_out/pmcd
