#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs: per-kernel durations and PMC counter averages."""
import glob
import json
import sqlite3
import sys


def summarize(d):
    out = {}
    for f in glob.glob(d + "/*.db"):
        c = sqlite3.connect(f)
        try:
            for r in c.execute("select name, count(*), avg(duration), min(duration), max(vgpr_count), max(sgpr_count), max(lds_size), "
                               "max(scratch_size) from kernels group by name"):
                if "jpgpu" in r[0]:
                    out.setdefault(r[0].split("(")[0], {}).update(calls=r[1], avg_us=round(r[2] / 1e3, 2), min_us=round(r[3] / 1e3, 2),
                                                                  vgpr=r[4], sgpr=r[5], lds=r[6], scratch=r[7])
        except sqlite3.Error as e:
            print("kernels:", e)
        try:
            cols = [x[1] for x in c.execute("pragma table_info('counters_collection')")]
            if cols:
                q = "select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"
                for k, n, v in c.execute(q):
                    if "jpgpu" in k:
                        out.setdefault(k.split("(")[0], {}).setdefault("pmc", {})[n] = round(v, 1)
        except sqlite3.Error as e:
            print("pmc:", e, cols)
    return out


def dispatches(dirs, substr, last):
    """Every dispatch of kernels whose name holds `substr`, in launch order, with its own duration and counters (the last `last`
    of them): kernels launched many times with very different work (the sync passes) do not average into anything."""
    rows = {}
    for d in dirs:
        for f in glob.glob(d + "/*.db"):
            c = sqlite3.connect(f)
            order = [r for r in c.execute("select dispatch_id, name, duration, grid_x, grid_y from kernels where name like ? order by start", ("%" + substr + "%",))]
            order = order[-last:]
            pos = {r[0]: i for i, r in enumerate(order)}
            for did, name, dur, gx, gy in order:
                rows.setdefault(pos[did], {}).setdefault("us", []).append(round(dur / 1e3, 1))
                rows[pos[did]]["grid"] = [gx, gy]
            try:
                for did, n, v in c.execute("select dispatch_id, counter_name, sum(value) from counters_collection where kernel_name like ? group by dispatch_id, counter_name",
                                           ("%" + substr + "%",)):
                    if did in pos:
                        rows[pos[did]][n] = round(v, 1)
            except sqlite3.Error as e:
                print("pmc:", e)
    return [dict(i=i, **rows[i]) for i in sorted(rows)]


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "--dispatches":  # --dispatches <kernel substr> <last N> dir...
        for r in dispatches(sys.argv[4:], sys.argv[2], int(sys.argv[3])):
            print(json.dumps(r))
        sys.exit(0)
    res = {}
    for d in sys.argv[1:]:
        for k, v in summarize(d).items():
            pm = v.pop("pmc", {})
            res.setdefault(k, {}).update(v)
            res[k].setdefault("pmc", {}).update(pm)
    print(json.dumps(res, indent=1))
