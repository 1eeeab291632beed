#!/usr/bin/env python3
"""A few jpgpu_pipeline_decode calls over N copies of a progressive file with every frame's scans decoded on the device
(JPGPU_PIPE_PROG_DEVICE_PERCENT=100): the command rocprofv3 wraps for the kernel trace / counter passes of the track walker
(huff_progw_kernel), and a quick look at call times.  --distinct: N different encoder-written frames instead of copies of one file."""
import argparse
import io
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=256)
ap.add_argument("--calls", type=int, default=5)
ap.add_argument("--percent", default="100")
ap.add_argument("--file", default="tests/golden/benches/tower_progressive.jpg")
ap.add_argument("--distinct", action="store_true")
args = ap.parse_args()
if args.percent != "auto":
    os.environ["JPGPU_PIPE_PROG_DEVICE_PERCENT"] = args.percent
R = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import jpeg_decoder_amd as J  # noqa: E402

J.process_init()
if args.distinct:
    from PIL import Image
    import synth
    files = []
    for k in range(min(args.images, 64)):
        buf = io.BytesIO()
        Image.fromarray(synth.synthetic_rgb(512, 512, seed=0x700 + k)).save(buf, format="JPEG", quality=85, subsampling="4:4:4", progressive=True)
        files.append(buf.getvalue())
    files = [files[i % len(files)] for i in range(args.images)]
else:
    files = [open(os.path.join(R, args.file), "rb").read()] * args.images
p = J.Pipeline()
for _ in range(args.calls):
    t0 = time.perf_counter()
    p.decode(files, device_entropy=True, download=False)
    t = p.timings()
    print("call ms", round((time.perf_counter() - t0) * 1e3, 3), "total_ms", round(t["total_ms"], 2), "on the device", t["images_device_progressive"], "walk+scan ms", round(t["dev_sync_ms"], 2), flush=True)
p.close()
