#!/bin/bash
# How many of a wave's 64 lanes work in a vector instruction of the device-entropy kernels (256 x 1080p as one sub-batch):
# SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU, per launch of huff_sync_pass_kernel in order (pass 0, 1, 2, ...).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/laneutil; rm -rf $O; mkdir -p $O
cat > /tmp/many.py <<PY
import io, os, sys, time
os.environ["JPGPU_PIPE_DEV_SUB"], os.environ["JPGPU_PIPE_MAX_DEV_SUBS"] = "256", "1"
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import jpeg_decoder_amd as J, synth
from PIL import Image
files = []
for i in range(4):
    buf = io.BytesIO(); Image.fromarray(synth.synthetic_rgb(1920, 1080, seed=0x5EED + i)).save(buf, format="JPEG", quality=85, subsampling="4:2:0"); files.append(buf.getvalue())
files = [files[i % 4] for i in range(256)]
p = J.Pipeline()
for _ in range(3):
    p.decode(files, device_entropy=True, download=False)
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $O/t -o p -- python /tmp/many.py > $O/log.txt 2>&1
cd $R
python - <<PY
import glob, sqlite3
f = glob.glob("$O/t/*.db")[0]
c = sqlite3.connect(f)
cols = [x[1] for x in c.execute("pragma table_info('counters_collection')")]
print(cols)
idc = "dispatch_id" if "dispatch_id" in cols else cols[0]
by = {}
for did, name, cn, val in c.execute(f"select {idc}, kernel_name, counter_name, sum(value) from counters_collection group by {idc}, kernel_name, counter_name"):
    by.setdefault((did, name), {})[cn] = val
for (did, name) in sorted(by)[-30:]:
    v = by[(did, name)]
    a = v.get("SQ_ACTIVE_INST_VALU", 0); t = v.get("SQ_THREAD_CYCLES_VALU", 0)
    print(did, name[:40].ljust(40), {k: int(x) for k, x in v.items()}, "lanes/instr %.1f" % (t / a if a else 0))
PY
