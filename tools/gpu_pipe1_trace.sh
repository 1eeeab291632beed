#!/bin/bash
# kernel trace of one 1080p image through the device-entropy pipeline (per-kernel durations and launch order)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pipe1; rm -rf $O; mkdir -p $O
cd /tmp
cat > /tmp/one.py <<PY
import io, os, sys
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import jpeg_decoder_amd as J, synth
from PIL import Image
buf = io.BytesIO(); Image.fromarray(synth.synthetic_rgb(1920, 1080, seed=1)).save(buf, format="JPEG", quality=85, subsampling="4:2:0")
p = J.Pipeline(threads=1)
for _ in range(12): p.decode([buf.getvalue()], device_entropy=True)
PY
timeout 300 rocprofv3 --kernel-trace -d $O/t -o k -- python /tmp/one.py > $O/log.txt 2>&1
cd $R
python - <<PY
import glob, sqlite3
f = glob.glob("$O/t/*.db")[0]
c = sqlite3.connect(f)
rows = list(c.execute("select name, start, end from kernels order by start"))
# the last decode: kernels after the last big gap
last = []
prev_end = None
groups = [[]]
for n, s, e in rows:
    if prev_end is not None and s - prev_end > 300000: groups.append([])
    groups[-1].append((n, s, e)); prev_end = e
g = groups[-1]
t0 = g[0][1]
print(len(groups), "groups; last group:", len(g), "kernels, span", (g[-1][2] - t0) / 1e3, "us")
for n, s, e in g:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {n[:90]}")
PY
