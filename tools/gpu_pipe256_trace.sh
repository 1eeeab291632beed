#!/bin/bash
# kernel trace of 256 x 1080p through the device-entropy pipeline: which kernels the call's GPU time is made of
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pipe256; rm -rf $O; mkdir -p $O
cd /tmp
cat > /tmp/many.py <<PY
import io, os, sys, time
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import jpeg_decoder_amd as J, synth
from PIL import Image
files = []
for i in range(8):
    buf = io.BytesIO(); Image.fromarray(synth.synthetic_rgb(1920, 1080, seed=i)).save(buf, format="JPEG", quality=85, subsampling="4:2:0"); files.append(buf.getvalue())
files = [files[i % 8] for i in range(256)]
p = J.Pipeline()
for _ in range(6):
    t0 = time.perf_counter(); p.decode(files, device_entropy=True, download=False); print("call ms", (time.perf_counter() - t0) * 1e3, p.timings(), flush=True)
PY
timeout 300 rocprofv3 --kernel-trace -d $O/t -o k -- python /tmp/many.py > $O/log.txt 2>&1
grep "call ms" $O/log.txt | tail -2 | cut -c1-300
cd $R
python - <<PY
import glob, sqlite3
f = glob.glob("$O/t/*.db")[0]
c = sqlite3.connect(f)
rows = list(c.execute("select name, start, end from kernels order by start"))
groups = [[]]; prev_end = None
for n, s, e in rows:
    if prev_end is not None and s - prev_end > 1500000: groups.append([])
    groups[-1].append((n, s, e)); prev_end = e
g = groups[-1]
t0 = g[0][1]
print(len(groups), "groups; last group:", len(g), "kernels, span", (g[-1][2] - t0) / 1e3, "us")
agg = {}
for n, s, e in g:
    k = n.split("(")[0][:60]; a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
for k, (cnt, us) in sorted(agg.items(), key=lambda x: -x[1][1]): print(f"{us:10.1f} us  x{cnt:3d}  {k}")
PY
