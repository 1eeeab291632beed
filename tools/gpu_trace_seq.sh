#!/bin/bash
# kernel-trace of a command, then the LAST n dispatches in order (name, duration): tools/gpu_trace_seq.sh <n> <cmd...>
N=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/trc
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trc -o t -- "$@" > $R/gpurun_out/trc.log 2>&1
cd $R
python - $N <<'PY'
import sqlite3, glob, sys
f = glob.glob('gpurun_out/trc/*.db')[0]
c = sqlite3.connect(f)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = list(c.execute("select name, start, end, duration from kernels order by start"))
rows = rows[-int(sys.argv[1]):]
t0 = rows[0][1]
for r in rows:
    print("%9.1f us  +%8.1f  %s" % ((r[1] - t0) / 1e3, r[3] / 1e3, r[0][:60]))
PY
