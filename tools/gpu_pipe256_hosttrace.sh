#!/bin/bash
# host-side trace (JPGPU_PIPE_TRACE) of a 256 x 1080p device-entropy pipeline call
R=$GRAFT_REPO_ROOT
JPGPU_PIPE_TRACE=1 python - 2>&1 <<PY | tail -40
import io, os, sys, time
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import jpeg_decoder_amd as J, synth
from PIL import Image
files = []
for i in range(8):
    buf = io.BytesIO(); Image.fromarray(synth.synthetic_rgb(1920, 1080, seed=i)).save(buf, format="JPEG", quality=85, subsampling="4:2:0"); files.append(buf.getvalue())
files = [files[i % 8] for i in range(256)]
p = J.Pipeline()
for _ in range(4):
    t0 = time.perf_counter(); p.decode(files, device_entropy=True, download=False); print("== call ms", (time.perf_counter() - t0) * 1e3, flush=True)
PY
