#!/bin/bash
# HBM traffic of the default workload (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) -> gpurun_out/pmc_traffic.json
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/profiles/round1/pmc_traffic.json $R/gpurun_out/pmc_traffic.json
for spec in "1080p-420:fused420:f420_" "1080p-422:fused422:f422_"; do
  IFS=: read wl path pat <<< "$spec"
  cd /tmp
  rm -rf $R/gpurun_out/tf_$wl $R/gpurun_out/tw_$wl
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/tf_$wl -o p -- python $R/bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/tw_$wl -o p -- python $R/bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
  cd $R/tools && python make_pmc_traffic.py $wl:$path ../gpurun_out/tf_$wl ../gpurun_out/tw_$wl ../gpurun_out/pmc_traffic.json $pat
done
