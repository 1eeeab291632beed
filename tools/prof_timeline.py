#!/usr/bin/env python3
"""Device timeline of the LAST burst of activity in a rocprofv3 --kernel-trace --memory-copy-trace database: every kernel and every
copy with its stream, start and end in ms relative to the burst's first operation (a burst: operations not separated by more than
`--gap` ms of idle device).  For looking at how a pipeline call's sub-batches — uploads, entropy kernels, pixel kernels — lie next to
one another:  python tools/prof_timeline.py <rocprof dir> [--gap 2] [--min-us 20]"""
import argparse
import glob
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--gap", type=float, default=2.0)
ap.add_argument("--min-us", type=float, default=20.0, help="leave out operations shorter than this (the listing, not the burst)")
ap.add_argument("--burst", type=int, default=-1, help="which burst (default: the last)")
args = ap.parse_args()
ops = []
for f in glob.glob(args.dir + "/*.db"):
    c = sqlite3.connect(f)
    for name, stream, s, e in c.execute("select name, stream_id, start, end from kernels"):
        ops.append((s, e, "k", str(stream), name.split("(")[0].replace("void ", "").replace("jpgpu::", "")[:48], 0))
    try:
        for name, stream, s, e, size in c.execute("select name, stream_id, start, end, size from memory_copies"):
            ops.append((s, e, "c", str(stream), name[:48], size))
    except sqlite3.Error:
        pass
ops.sort()
bursts, cur, last_end = [], [], None
for o in ops:
    if last_end is not None and o[0] - last_end > args.gap * 1e6:
        bursts.append(cur)
        cur = []
    cur.append(o)
    last_end = o[1] if last_end is None else max(last_end, o[1])
if cur:
    bursts.append(cur)
print(len(bursts), "bursts; operations per burst:", [len(b) for b in bursts][-12:])
b = bursts[args.burst]
t0 = b[0][0]
print("burst of %d operations, %.3f ms" % (len(b), (max(o[1] for o in b) - t0) / 1e6))
for s, e, kind, stream, name, size in b:
    if (e - s) / 1e3 < args.min_us:
        continue
    print("%8.3f .. %8.3f  %7.1f us  stream %-4s %s %s%s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, stream, "copy  " if kind == "c" else "kernel", name,
                                                            "  %.1f MB %.1f GB/s" % (size / 1e6, size / (e - s)) if kind == "c" and e > s else ""))
