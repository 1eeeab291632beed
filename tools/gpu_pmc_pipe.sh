#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the device-entropy pipeline's kernels (256 x 1080p as one sub-batch): tools/gpu_pmc_pipe.sh <tag> [ENV=VAL ...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift; O=$R/gpurun_out/pmcpipe_$tag; rm -rf $O; mkdir -p $O
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
for _ in range(4):
    p.decode(files, device_entropy=True, download=False)
PY
cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p -- python /tmp/many.py > $O/fetch.log 2>&1
env "$@" timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p -- python /tmp/many.py > $O/write.log 2>&1
cd $R
python tools/prof_summary.py $O/fetch $O/write | python -c "
import sys, json
d = json.load(sys.stdin)
for k, v in d.items():
    p = v.get('pmc', {})
    if 'FETCH_SIZE' in p: print('$tag %-40s calls %3d  fetch x2 %8.1f MB  write %8.1f MB per launch' % (k[:40], v['calls'], p['FETCH_SIZE'] * 2 * 1024 / 1e6, p.get('WRITE_SIZE', 0) * 1024 / 1e6))"
