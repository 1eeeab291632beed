"""Shared helpers for the reftest-style comparisons (mirror of the reference's
tests/reftest/mod.rs:27-164 and tests/common/mod.rs:6-40, restated in Python)."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REFTEST = os.path.join(GOLDEN, "reftest")


def disabled_list():
    out = []
    for line in open(os.path.join(REFTEST, "disabled.list")):
        line = line.strip()
        if line and not line.startswith("#"):
            out.append(line)
    return out


def reftest_files(include_disabled=False):
    """tests/common/mod.rs:6-40 — every *.jpg / *.jpeg minus disabled.list."""
    files = sorted(glob.glob(os.path.join(REFTEST, "*.jp*g")) + glob.glob(os.path.join(REFTEST, "mozilla", "*.jp*g")))
    rel = [os.path.relpath(f, REFTEST) for f in files]
    if not include_disabled:
        dis = set(disabled_list())
        rel = [r for r in rel if r not in dis]
    return rel


def golden_hashes():
    return json.load(open(os.path.join(GOLDEN, "decode_sha256.json")))


def cmyk_to_rgb(data):
    """tests/reftest/mod.rs:138-164 (f32 arithmetic, truncating casts)."""
    p = np.asarray(data, dtype=np.uint8).reshape(-1, 4).astype(np.float32) / np.float32(255.0)
    c, m, y, k = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
    one = np.float32(1.0)
    c = c * (one - k) + k
    m = m * (one - k) + k
    y = y * (one - k) + k
    rgb = np.stack([(one - c) * np.float32(255.0), (one - m) * np.float32(255.0), (one - y) * np.float32(255.0)], 1)
    return rgb.astype(np.uint8).reshape(-1)


def load_png(path):
    """PNG expectation as the reference's harness sees it through png-0.16 defaults
    (16-bit stripped to the high byte, 1-bit expanded to 0/255, alpha dropped)."""
    from PIL import Image

    im = Image.open(path)
    if im.mode in ("I;16", "I;16B", "I"):
        a = (np.array(im).astype(np.uint32) >> 8).astype(np.uint8)
    elif im.mode == "1":
        a = np.array(im).astype(np.uint8) * 255
    elif im.mode == "RGBA":
        a = np.array(im)[..., :3]
    elif im.mode == "P":
        a = np.array(im.convert("RGB"))
    else:
        a = np.array(im)
    return a.reshape(-1), im.size


def max_diff_vs_png(pixels, ncomp, png_path):
    ref, _ = load_png(png_path)
    got = cmyk_to_rgb(pixels) if ncomp == 4 else np.asarray(pixels, dtype=np.uint8).reshape(-1)
    assert ref.size == got.size, (ref.size, got.size)
    return int(np.abs(ref.astype(np.int32) - got.astype(np.int32)).max())


ANCHOR = os.path.join(GOLDEN, "anchor")


def anchor_files():
    """4:4:0 / 4:1:1 files with libjpeg-turbo's decodes beside them (tests/golden/anchor/README.md): the external anchor for
    the two upsamplers the reference holds no fixture for."""
    return sorted(os.path.basename(f) for f in glob.glob(os.path.join(ANCHOR, "*.jpg")))
