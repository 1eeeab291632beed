#!/bin/bash
# kernel-trace of a command; for the LAST `n` dispatches: per kernel name calls / sum / avg, the busy union and the span: tools/gpu_trace_span.sh <n> <cmd...>
N=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/trc
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/trc -o t -- "$@" > $R/gpurun_out/trc.log 2>&1
cd $R
python - $N <<'PY'
import sqlite3, glob, sys, collections
f = glob.glob('gpurun_out/trc/*.db')[0]
c = sqlite3.connect(f)
rows = list(c.execute("select name, start, end, duration from kernels order by start"))
rows = rows[-int(sys.argv[1]):]
t0, t1 = rows[0][1], max(r[2] for r in rows)
agg = collections.OrderedDict()
for r in rows:
    a = agg.setdefault(r[0][:50], [0, 0.0])
    a[0] += 1; a[1] += r[3] / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-52s calls %5d sum %10.1f us avg %8.1f" % (k, v[0], v[1], v[1] / v[0]))
# union of busy intervals
iv = sorted((r[1], r[2]) for r in rows)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print("span %.1f us, some kernel running %.1f us, sum of durations %.1f us" % ((t1 - t0) / 1e3, busy / 1e3, sum(r[3] for r in rows) / 1e3))
try:
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view') and name like '%memory_cop%'")]
    for t in tabs[:1]:
        cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
        print(t, cols)
        m = list(c.execute("select * from %s" % t))
        print(len(m), "copies")
except Exception as e:
    print("copies:", e)
PY
