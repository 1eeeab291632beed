#!/bin/bash
# kernel-trace of an arbitrary command: tools/gpu_trace_cmd.sh <cmd...>
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/trc
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trc -o t -- "$@" > $R/gpurun_out/trc.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob
f = glob.glob('gpurun_out/trc/*.db')[0]
c = sqlite3.connect(f)
for r in c.execute("select name, count(*), avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, max(vgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by sum(duration) desc"):
    print("%-70s calls %4d avg %9.1f us min %9.1f max %9.1f vgpr %s lds %s scratch %s" % (r[0][:70], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
PY
