#!/usr/bin/env python3
"""Build profiles/<round>/pmc_traffic.json from rocprofv3 PMC passes (tools/profile_round.sh):
   make_pmc_traffic.py <workload:path> <fetch_dir> <write_dir> <out.json> [kernel-name-substring ...]
HBM bytes per decode = sum over the decode's kernels of FETCH_SIZE*2 (gfx950 correction, MI355X_MICROARCH.md §HBM;
counters are KB) + WRITE_SIZE, each averaged per dispatch."""
import json
import sys

import kernel_sources
from prof_summary import summarize


def main():
    key, fetch_dir, write_dir, out = sys.argv[1:5]
    wanted = sys.argv[5:]
    f, w = summarize(fetch_dir), summarize(write_dir)
    per, total = {}, 0
    for name in sorted(set(f) | set(w)):
        if wanted and not any(s in name for s in wanted):
            continue
        fk = f.get(name, {}).get("pmc", {}).get("FETCH_SIZE")
        wk = w.get(name, {}).get("pmc", {}).get("WRITE_SIZE")
        if fk is None or wk is None:
            continue
        b = int(fk * 1024 * 2 + wk * 1024)
        per[name] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes": b}
        total += b
    try:
        doc = json.load(open(out))
    except (OSError, ValueError):
        doc = {}
    doc[key] = {"hbm_bytes_per_decode": total, "per_kernel": per, "commit": kernel_sources.head_commit(),
                "pixel_kernel_sources_sha256": kernel_sources.sha256(),
                "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs; FETCH_SIZE doubled (gfx950: 128-B "
                          "requests tallied at 64 B, MI355X_MICROARCH.md §HBM); counters are KB; average per dispatch; sum over "
                          "the kernels of one decode of the batch"}
    json.dump(doc, open(out, "w"), indent=1)
    print(key, total)


if __name__ == "__main__":
    main()
