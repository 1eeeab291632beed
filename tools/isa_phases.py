#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing, split at s_barrier:
   python tools/isa_phases.py fused.s f420_main_kernelILi2ELj256
Counts are static (loops are counted once); good enough to compare phases of straight-line kernels."""
import re
import sys
from collections import Counter


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    phase, phases = Counter(), []
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_barrier":
            phases.append(phase)
            phase = Counter()
            continue
        cls = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else
               "vmem" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else "other")
        phase[cls] += 1
        phase["op:" + op] += 1
    phases.append(phase)
    for i, p in enumerate(phases):
        print(f"phase {i}: " + " ".join(f"{k}={p[k]}" for k in ("valu", "salu", "lds", "vmem", "other")))
        top = sorted(((v, k[3:]) for k, v in p.items() if k.startswith("op:v_")), reverse=True)[:14]
        print("   " + ", ".join(f"{k}:{v}" for v, k in top))


if __name__ == "__main__":
    main()
