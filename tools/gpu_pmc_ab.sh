#!/bin/bash
# PMC passes for a 4:2:0 A/B: two-pass (default) vs the single-launch strip walk (JPGPU_420_STRIP=1).  Separate --pmc runs
# (no trace domains besides the kernel trace), summaries merged by tools/prof_summary.py.
#   usage: bash tools/gpu_pmc_ab.sh <out-subdir> [extra env assignments for the strip arm]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_ab}
shift
mkdir -p $O
cd /tmp
BENCH="python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-classes"
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
P2="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
for arm in twopass strip; do
  if [ $arm = strip ]; then export JPGPU_420_STRIP=1 "$@"; fi
  timeout 200 rocprofv3 --kernel-trace --pmc $P1 -d $O/$arm-p1 -o p -- $BENCH > $O/$arm-p1.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc $P2 -d $O/$arm-p2 -o p -- $BENCH > $O/$arm-p2.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/$arm-f -o p -- $BENCH > $O/$arm-f.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/$arm-w -o p -- $BENCH > $O/$arm-w.log 2>&1
  python $R/tools/prof_summary.py $O/$arm-p1 $O/$arm-p2 $O/$arm-f $O/$arm-w > $O/$arm.json 2> $O/$arm.err
  rm -rf $O/$arm-p1 $O/$arm-p2 $O/$arm-f $O/$arm-w
done
cat $O/twopass.json $O/strip.json
