#!/usr/bin/env python3
"""A small deterministic baseline JPEG encoder (numpy only) for the bench's and the tests' end-to-end inputs.

Why it exists: bench.py's `e2e` block decodes JPEG FILES, and SURVEY §8d asks that nothing which gates a number depend
on an encoder that may be absent on the GPU box (Pillow is used when importable — libjpeg-turbo, the encoder real files
come from — and this one otherwise, and bench.py says which).  It writes what encoders write by default: JFIF, 8-bit
baseline (SOF0), one interleaved scan, the Annex K Huffman tables, libjpeg's quality scaling of the Annex K quantization
tables, optional restart intervals.  The forward path (colour transform, box subsampling, FDCT, quantisation) is
tests/synth.py's, i.e. the very coefficients the kernel-only bench uses, so a file written here decodes to exactly the
pixels of that workload.

    python tools/baseline_encoder.py out.jpg --width 1920 --height 1080 --sampling 420 --quality 85
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# zig-zag position k -> natural (row-major) index (ITU T.81 figure A.6)
UNZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35,
                     42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])

# ITU T.81 Annex K.3 typical Huffman tables: (BITS[1..16], HUFFVAL)
_DC_L = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
_DC_C = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
_AC_L = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125],
         [1, 2, 3, 0, 4, 17, 5, 18, 33, 49, 65, 6, 19, 81, 97, 7, 34, 113, 20, 50, 129, 145, 161, 8, 35, 66, 177, 193, 21, 82, 209, 240, 36, 51, 98,
          114, 130, 9, 10, 22, 23, 24, 25, 26, 37, 38, 39, 40, 41, 42, 52, 53, 54, 55, 56, 57, 58, 67, 68, 69, 70, 71, 72, 73, 74, 83, 84, 85, 86, 87,
          88, 89, 90, 99, 100, 101, 102, 103, 104, 105, 106, 115, 116, 117, 118, 119, 120, 121, 122, 131, 132, 133, 134, 135, 136, 137, 138, 146, 147,
          148, 149, 150, 151, 152, 153, 154, 162, 163, 164, 165, 166, 167, 168, 169, 170, 178, 179, 180, 181, 182, 183, 184, 185, 186, 194, 195, 196,
          197, 198, 199, 200, 201, 202, 210, 211, 212, 213, 214, 215, 216, 217, 218, 225, 226, 227, 228, 229, 230, 231, 232, 233, 234, 241, 242, 243,
          244, 245, 246, 247, 248, 249, 250])
_AC_C = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119],
         [0, 1, 2, 3, 17, 4, 5, 33, 49, 6, 18, 65, 81, 7, 97, 113, 19, 34, 50, 129, 8, 20, 66, 145, 161, 177, 193, 9, 35, 51, 82, 240, 21, 98, 114,
          209, 10, 22, 36, 52, 225, 37, 241, 23, 24, 25, 26, 38, 39, 40, 41, 42, 53, 54, 55, 56, 57, 58, 67, 68, 69, 70, 71, 72, 73, 74, 83, 84, 85,
          86, 87, 88, 89, 90, 99, 100, 101, 102, 103, 104, 105, 106, 115, 116, 117, 118, 119, 120, 121, 122, 130, 131, 132, 133, 134, 135, 136, 137,
          138, 146, 147, 148, 149, 150, 151, 152, 153, 154, 162, 163, 164, 165, 166, 167, 168, 169, 170, 178, 179, 180, 181, 182, 183, 184, 185, 186,
          194, 195, 196, 197, 198, 199, 200, 201, 202, 210, 211, 212, 213, 214, 215, 216, 217, 218, 226, 227, 228, 229, 230, 231, 232, 233, 234, 242,
          243, 244, 245, 246, 247, 248, 249, 250])


def _code_table(spec):
    """(BITS, HUFFVAL) -> code[256], length[256] (T.81 Annex C)."""
    bits, vals = spec
    code = np.zeros(256, np.int64)
    length = np.zeros(256, np.int64)
    c, k = 0, 0
    for ln in range(1, 17):
        for _ in range(bits[ln - 1]):
            code[vals[k]], length[vals[k]] = c, ln
            c += 1
            k += 1
        c <<= 1
    return code, length


_TABLES = {"dc": [_code_table(_DC_L), _code_table(_DC_C)], "ac": [_code_table(_AC_L), _code_table(_AC_C)]}


def _category(v):
    """number of bits of |v| (0 for 0)"""
    a = np.abs(v).astype(np.int64)
    cat = np.zeros(a.shape, np.int64)
    nz = a > 0
    cat[nz] = np.floor(np.log2(a[nz])).astype(np.int64) + 1
    return cat


def _amplitude_bits(v, cat):
    """T.81 F.1.2.1: v >= 0 -> v, else v + 2^cat - 1 (the low `cat` bits of v - 1)"""
    v = v.astype(np.int64)
    return np.where(v >= 0, v, v + (np.int64(1) << cat) - 1)


def _pack_bits(values, lengths):
    """entries (value, length <= 32) in order -> bytes, last byte padded with 1 bits (T.81 F.1.2.3), 0xFF stuffed"""
    lengths = lengths.astype(np.int64)
    off = np.concatenate(([0], np.cumsum(lengths)[:-1]))
    total = int(off[-1] + lengths[-1]) if len(lengths) else 0
    nbytes = (total + 7) // 8
    out = np.zeros(nbytes + 8, np.uint32)
    b, sh = off >> 3, off & 7
    w = values.astype(np.uint64) << (np.uint64(64) - lengths.astype(np.uint64) - sh.astype(np.uint64))  # left-aligned in a 64-bit window at byte b
    for k in range(5):  # 7 + 32 bits span at most five bytes
        np.add.at(out, b + k, ((w >> np.uint64(56 - 8 * k)) & np.uint64(0xFF)).astype(np.uint32))
    out = out[:nbytes].astype(np.uint8)
    if total & 7:
        out[-1] |= (1 << (8 - (total & 7))) - 1
    ff = np.flatnonzero(out == 0xFF)
    return np.insert(out, ff + 1, 0).tobytes()


def _entropy_segment(blocks, tab, comp_of, pred, tables=None):
    """blocks: (n, 64) zig-zag order int, in stream order; tab: (n,) 0 luma / 1 chroma tables; comp_of: (n,) component (DC
    predictors); pred: dict component -> predictor at the start of the segment.  -> entropy-coded bytes."""
    _T = tables or _TABLES
    n = blocks.shape[0]
    dc = blocks[:, 0].astype(np.int64)
    diff = np.zeros(n, np.int64)
    for c in np.unique(comp_of):
        m = comp_of == c
        d = dc[m]
        diff[m] = d - np.concatenate(([pred.get(int(c), 0)], d[:-1]))
        pred[int(c)] = int(d[-1])
    KEY = 65 * 4
    blk = np.arange(n, dtype=np.int64)
    keys, vals, lens = [], [], []
    # DC
    cat = _category(diff)
    for t in (0, 1):
        m = tab == t
        code, ln = _T["dc"][t]
        keys.append(blk[m] * KEY + 3)
        vals.append((code[cat[m]] << cat[m]) | _amplitude_bits(diff[m], cat[m]))
        lens.append(ln[cat[m]] + cat[m])
    # AC
    ac = blocks[:, 1:].astype(np.int64)
    bi, pi = np.nonzero(ac)           # row-major: ascending position within a block
    pos = pi + 1
    first = np.concatenate(([True], bi[1:] != bi[:-1]))
    prev = np.where(first, 0, np.concatenate(([0], pos[:-1])))
    run = pos - prev - 1
    v = ac[bi, pi]
    cat = _category(v)
    tt = tab[bi]
    for t in (0, 1):
        m = tt == t
        code, ln = _T["ac"][t]
        sym = ((run[m] & 15) << 4) | cat[m]
        keys.append(bi[m] * KEY + pos[m] * 4 + 3)
        vals.append((code[sym] << cat[m]) | _amplitude_bits(v[m], cat[m]))
        lens.append(ln[sym] + cat[m])
        for j in range(3):  # ZRL symbols in front of a coefficient whose run reaches 16 / 32 / 48
            mz = m & (run >= 16 * (j + 1))
            keys.append(bi[mz] * KEY + pos[mz] * 4 + j)
            vals.append(np.full(int(mz.sum()), code[0xF0]))
            lens.append(np.full(int(mz.sum()), ln[0xF0]))
    # EOB unless the last coefficient of the block is non-zero
    last = np.zeros(n, np.int64)
    np.maximum.at(last, bi, pos)
    for t in (0, 1):
        m = (tab == t) & (last < 63)
        code, ln = _T["ac"][t]
        keys.append(blk[m] * KEY + 64 * 4)
        vals.append(np.full(int(m.sum()), code[0]))
        lens.append(np.full(int(m.sum()), ln[0]))
    keys, vals, lens = np.concatenate(keys), np.concatenate(vals), np.concatenate(lens)
    order = np.argsort(keys, kind="stable")
    return _pack_bits(vals[order], lens[order])


def _marker(m, payload=b""):
    return bytes([0xFF, m]) + (len(payload) + 2).to_bytes(2, "big") + bytes(payload)


def encode_from_coefficients(comps, qts, coefs, width, height, restart_interval=0, identifiers=None, huffman=None):
    """comps: components with horizontal_sampling_factor / vertical_sampling_factor / block_width / block_height (the geometry of
    src/parser.rs:282-310, e.g. jpeg_decoder_amd.make_components); qts: per component, 64 values <= 255 in natural order;
    coefs: per component, block-raster natural-order int16 (the Worker layout).  1 component -> grayscale, 3 -> YCbCr (JFIF).
    Components whose table equals the first one's share DQT 0 and the luminance Huffman tables; the others use the second set.
    huffman: optional {"dc": [(BITS, HUFFVAL), (BITS, HUFFVAL)], "ac": [...]} instead of Annex K's tables (every symbol the data needs
    must have a code; tests use it for streams with the shortest codes a table can have)."""
    specs = huffman or {"dc": [_DC_L, _DC_C], "ac": [_AC_L, _AC_C]}
    tables = {k: [_code_table(sp) for sp in v] for k, v in specs.items()}
    ncomp = len(comps)
    assert ncomp in (1, 3)
    H = [int(c.horizontal_sampling_factor) for c in comps]
    V = [int(c.vertical_sampling_factor) for c in comps]
    if ncomp == 1:
        H, V = [1], [1]  # a single-component scan is not interleaved: one block per MCU whatever the declared factors
    bw = [int(c.block_width) for c in comps]
    bh = [int(c.block_height) for c in comps]
    mcu_w, mcu_h = bw[0] // H[0], bh[0] // V[0]
    qts = [np.asarray(q, np.int64).reshape(64) for q in qts]
    tq = [0 if np.array_equal(q, qts[0]) else 1 for q in qts]
    assert all(int(q.max()) <= 255 and int(q.min()) >= 1 for q in qts)
    # blocks in stream order
    zz = []
    for c in range(ncomp):
        a = np.asarray(coefs[c], np.int64).reshape(bh[c], bw[c], 64)[:, :, UNZIGZAG]
        a[..., 1:] = np.clip(a[..., 1:], -1023, 1023)  # baseline: AC categories 1..10
        zz.append(a)
    per_mcu = []
    for c in range(ncomp):
        for v in range(V[c]):
            for h in range(H[c]):
                per_mcu.append((c, zz[c][v::V[c], h::H[c]][:mcu_h, :mcu_w]))  # (mcu_h, mcu_w, 64)
    stream = np.stack([a for _c, a in per_mcu], axis=2).reshape(mcu_h * mcu_w, len(per_mcu), 64)
    comp_of_q = np.array([c for c, _a in per_mcu])
    n_mcu = mcu_h * mcu_w
    ri = int(restart_interval)
    pred = {}
    parts = []
    seg = ri if ri else n_mcu
    for k, m0 in enumerate(range(0, n_mcu, seg)):
        m1 = min(n_mcu, m0 + seg)
        blocks = stream[m0:m1].reshape(-1, 64)
        comp_of = np.tile(comp_of_q, m1 - m0)
        if ri:
            pred = {}
        parts.append(_entropy_segment(blocks, np.array(tq)[comp_of], comp_of, pred, tables))
        if ri and m1 < n_mcu:
            parts.append(bytes([0xFF, 0xD0 + (k & 7)]))
    ids = list(identifiers) if identifiers else [1, 2, 3][:ncomp]
    out = [b"\xff\xd8", _marker(0xE0, b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0")]
    for t in sorted(set(tq)):
        q = qts[tq.index(t)]
        out.append(_marker(0xDB, bytes([t]) + bytes(int(x) for x in q[UNZIGZAG])))
    out.append(_marker(0xC0, bytes([8]) + int(height).to_bytes(2, "big") + int(width).to_bytes(2, "big") + bytes([ncomp]) +
                       b"".join(bytes([ids[c], (int(comps[c].horizontal_sampling_factor) << 4) | int(comps[c].vertical_sampling_factor), tq[c]]) for c in range(ncomp))))
    for t in sorted(set(tq)):
        for cls, spec in ((0, specs["dc"][t]), (1, specs["ac"][t])):
            out.append(_marker(0xC4, bytes([(cls << 4) | t]) + bytes(spec[0]) + bytes(spec[1])))
    if ri:
        out.append(_marker(0xDD, ri.to_bytes(2, "big")))
    out.append(_marker(0xDA, bytes([ncomp]) + b"".join(bytes([ids[c], (tq[c] << 4) | tq[c]]) for c in range(ncomp)) + b"\0\x3f\0"))
    out += parts
    out.append(b"\xff\xd9")
    return b"".join(out)


SAMPLINGS = {"420": [(2, 2), (1, 1), (1, 1)], "422": [(2, 1), (1, 1), (1, 1)], "444": [(1, 1), (1, 1), (1, 1)], "440": [(1, 2), (1, 1), (1, 1)],
             "411": [(4, 1), (1, 1), (1, 1)], "gray": [(1, 1)]}


def encode_rgb(rgb, quality=85, sampling="420", restart_interval=0):
    """8-bit RGB array (H, W, 3) -> baseline JFIF bytes (colour transform, box subsampling, FDCT and quantisation: tests/synth.py)."""
    import jpeg_decoder_amd as J
    import synth
    h, w = rgb.shape[:2]
    samp = SAMPLINGS[sampling]
    comps, _mcu = J.make_components(w, h, samp)
    lum, chr_ = synth.quality_tables(quality)
    qts = [lum, chr_, chr_][: len(samp)]
    coefs = synth.coefficients_from_rgb(rgb, comps, "gray" if sampling == "gray" else "ycbcr", qts)
    return encode_from_coefficients(list(comps), qts, coefs, w, h, restart_interval)


def synthetic_jpeg(width, height, quality=85, sampling="420", seed=0x5EED, restart_interval=0):
    """The bench's synthetic image (tests/synth.py::synthetic_rgb) as a baseline JPEG."""
    import synth
    return encode_rgb(synth.synthetic_rgb(width, height, seed=seed), quality, sampling, restart_interval)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--quality", type=int, default=85)
    ap.add_argument("--sampling", default="420", choices=sorted(SAMPLINGS))
    ap.add_argument("--restart-interval", type=int, default=0)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    a = ap.parse_args()
    data = synthetic_jpeg(a.width, a.height, a.quality, a.sampling, a.seed, a.restart_interval)
    open(a.out, "wb").write(data)
    print(f"{a.out}: {len(data)} bytes")


if __name__ == "__main__":
    main()
