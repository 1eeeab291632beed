#!/usr/bin/env python3
"""A few jpgpu_pipeline_decode calls over N copies of the bench's 1080p 4:2:0 files, entropy decoding on the device: the command
rocprofv3 wraps for the kernel trace / counter passes of tools/gpu.sh (pipe256), and a quick look at call times per knob setting.
  --one-sub-batch  the whole call as ONE sub-batch (kernels one after the other: what the phase figures of bench.py's
                   e2e.kernels_256_one_sub_batch time)"""
import argparse
import io
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=256)
ap.add_argument("--calls", type=int, default=6)
ap.add_argument("--one-sub-batch", action="store_true")
ap.add_argument("--restart-rows", type=int, default=0)
args = ap.parse_args()
if args.one_sub_batch:
    os.environ["JPGPU_PIPE_DEV_SUB"], os.environ["JPGPU_PIPE_MAX_DEV_SUBS"] = str(args.images), "1"
R = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import jpeg_decoder_amd as J  # noqa: E402

J.process_init()
import bench  # noqa: E402
import synth  # noqa: E402

distinct, who = bench.e2e_files(synth, 1920, 1080, "auto", restart_rows=args.restart_rows)
files = [distinct[i % len(distinct)] for i in range(args.images)]
p = J.Pipeline()
ts = []
for _ in range(args.calls):
    t0 = time.perf_counter()
    p.decode(files, device_entropy=True, download=False)
    print("call ms", round((time.perf_counter() - t0) * 1e3, 3), flush=True)
    ts.append(round(p.timings()["total_ms"], 2))
print("total_ms per call", ts, "| files by", who)
p.close()
