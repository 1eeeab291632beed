#!/usr/bin/env python3
"""Overlap of the kernels of a traced run (rocprofv3 --kernel-trace rocpd database): over the LAST `window_ms` of GPU activity, per kernel
name the summed duration and the time during which at least one dispatch of it was running, the union over all kernels, and the mean
number of dispatches in flight.  python tools/trace_overlap.py <dir with *.db> [window_ms]"""
import glob
import sqlite3
import sys


def main(d, window_ms=50.0):
    for f in glob.glob(d + "/*.db"):
        c = sqlite3.connect(f)
        rows = [(n.split("(")[0], s, e) for n, s, e in c.execute("select name, start, end from kernels order by start") if "jpgpu" in n]
        if not rows:
            continue
        t_end = max(e for _, _, e in rows)
        t0 = t_end - window_ms * 1e6
        rows = [(n, max(s, t0), e) for n, s, e in rows if e > t0]
        def union(iv):
            iv = sorted(iv)
            tot, cur_s, cur_e = 0, None, None
            for s, e in iv:
                if cur_e is None or s > cur_e:
                    if cur_e is not None:
                        tot += cur_e - cur_s
                    cur_s, cur_e = s, e
                else:
                    cur_e = max(cur_e, e)
            return tot + (cur_e - cur_s if cur_e is not None else 0)
        names = sorted({n for n, _, _ in rows})
        allu = union([(s, e) for _, s, e in rows])
        print(f"window {window_ms} ms: GPU busy (any jpgpu kernel) {allu / 1e6:.2f} ms; sum of durations {sum(e - s for _, s, e in rows) / 1e6:.2f} ms; "
              f"mean dispatches in flight {sum(e - s for _, s, e in rows) / max(allu, 1):.2f}")
        for n in names:
            iv = [(s, e) for m, s, e in rows if m == n]
            print(f"  {n[:60]:60s} dispatches {len(iv):5d}  sum {sum(e - s for s, e in iv) / 1e6:8.2f} ms  running {union(iv) / 1e6:7.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 50.0)
