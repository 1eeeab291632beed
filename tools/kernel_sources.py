#!/usr/bin/env python3
"""Which sources the pixel kernels are built from, and one sha256 over them: profiles/roundN/pmc_traffic.json records it next to the
counter values (tools/make_pmc_traffic.py), bench.py reports whether the figure it quotes was taken on the sources it runs, and
tests/test_profiles.py fails when the newest round's headline entry was not (evidence on the final code)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERNS = ["fused*.hpp", "fused.hip", "kernels.hip", "kernels.hpp", "pixel_math.hpp", "idct_plane_body.hpp", "upsample_color_body.hpp", "jobs.hpp", "range_stats.hpp"]


def files():
    csrc = os.path.join(ROOT, "jpeg-decoder_amd", "csrc")
    out = []
    for p in PATTERNS:
        out += glob.glob(os.path.join(csrc, p))
    return sorted(set(out))


def sha256():
    h = hashlib.sha256()
    for f in files():
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def head_commit():
    """The commit the tree was at when the GPU call was launched (tools/.head_commit, written by the launching shell: the GPU box has
    no .git), or git's own answer here."""
    try:
        return open(os.path.join(ROOT, "tools", ".head_commit")).read().strip()
    except OSError:
        pass
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:  # noqa: BLE001
        return None


if __name__ == "__main__":
    print(sha256(), head_commit())
