#!/usr/bin/env python3
"""A small progressive JPEG encoder (numpy + pure Python) with an arbitrary SCAN SCRIPT — what `jpegtran -scans` does, which this image
does not have.  Test infrastructure: the device decoder for progressive frames (csrc/huff_prog_wave.hpp) and the host front-end are
only ever fed libjpeg's default script by Pillow; scripts that split a band's refinement (Y 1-5 | Y 6-63 | refine 1-5 | refine 6-63:
two waves OR-ing into the same mask words), refine DC before AC, interleave subsets of components, or cut bands in odd
places exist in the wild and need files.

The entropy coder follows ITU T.81 G.1.2 as libjpeg implements it (jcphuff.c: end-of-band runs, correction bits buffered behind the
symbol they follow), with an optimal Huffman table per scan (T.81 K.2, lengths limited to 16).  The forward path (colour transform,
subsampling, FDCT, quantisation) is tests/synth.py's.

    script: list of scans, each (components, ss, se, ah, al) — components: indices into the frame's components.
    DEFAULT_SCRIPT_YCC is libjpeg's; SPLIT_REFINEMENT_YCC refines Y's two bands separately.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
from baseline_encoder import SAMPLINGS, UNZIGZAG, _marker  # noqa: E402

DEFAULT_SCRIPT_YCC = [((0, 1, 2), 0, 0, 0, 1), ((0,), 1, 5, 0, 2), ((2,), 1, 63, 0, 1), ((1,), 1, 63, 0, 1), ((0,), 6, 63, 0, 2),
                      ((0,), 1, 63, 2, 1), ((0, 1, 2), 0, 0, 1, 0), ((2,), 1, 63, 1, 0), ((1,), 1, 63, 1, 0), ((0,), 1, 63, 1, 0)]
# Y's band cut in two and every part refined on its own, the refinement of 1-5 in front of the first scan of 6-63 of the NEXT bit plane,
# chroma in three bands, DC refined early
SPLIT_REFINEMENT_YCC = [((0, 1, 2), 0, 0, 0, 1), ((0,), 1, 5, 0, 2), ((0,), 6, 63, 0, 2), ((0,), 1, 5, 2, 1), ((0, 1, 2), 0, 0, 1, 0),
                        ((0,), 6, 63, 2, 1), ((1,), 1, 2, 0, 0), ((2,), 1, 9, 0, 1), ((1,), 3, 20, 0, 0), ((0,), 1, 5, 1, 0), ((2,), 10, 63, 0, 1),
                        ((1,), 21, 63, 0, 0), ((0,), 6, 63, 1, 0), ((2,), 1, 9, 1, 0), ((2,), 10, 63, 1, 0)]
SPLIT_REFINEMENT_GRAY = [((0,), 0, 0, 0, 0), ((0,), 1, 3, 0, 2), ((0,), 4, 63, 0, 2), ((0,), 4, 63, 2, 1), ((0,), 1, 3, 2, 1), ((0,), 1, 3, 1, 0), ((0,), 4, 63, 1, 0)]


class _Bits:
    """Tokens of a scan: ("sym", symbol) | ("bits", value, n); written once the table is known."""

    def __init__(self):
        self.tok = []

    def sym(self, s):
        self.tok.append((0, int(s), 0))

    def bits(self, v, n):
        if n:
            self.tok.append((1, int(v) & ((1 << n) - 1), int(n)))


def _optimal_table(freq):
    """T.81 K.2 / jpeg_gen_optimal_table: symbol frequencies -> (BITS[16], HUFFVAL), no code longer than 16 bits, none all ones."""
    freq = list(freq) + [1]  # the reserved code point
    codesize = [0] * 257
    others = [-1] * 257
    while True:
        c1, v = -1, 1 << 60
        for i in range(257):
            if freq[i] and freq[i] <= v:
                v, c1 = freq[i], i
        c2, v = -1, 1 << 60
        for i in range(257):
            if freq[i] and freq[i] <= v and i != c1:
                v, c2 = freq[i], i
        if c2 < 0:
            break
        freq[c1] += freq[c2]
        freq[c2] = 0
        codesize[c1] += 1
        while others[c1] >= 0:
            c1 = others[c1]
            codesize[c1] += 1
        others[c1] = c2
        codesize[c2] += 1
        while others[c2] >= 0:
            c2 = others[c2]
            codesize[c2] += 1
    bits = [0] * 33
    for i in range(257):
        if codesize[i]:
            bits[codesize[i]] += 1
    for i in range(32, 16, -1):
        while bits[i] > 0:
            j = i - 2
            while bits[j] == 0:
                j -= 1
            bits[i] -= 2
            bits[i - 1] += 1
            bits[j + 1] += 2
            bits[j] -= 1
    i = 16
    while bits[i] == 0:
        i -= 1
    bits[i] -= 1  # the reserved code point
    vals = [s for ln in range(1, 33) for s in range(256) if codesize[s] == ln]
    return bits[1:17], vals


def _codes(bits, vals):
    code, length, c, k = {}, {}, 0, 0
    for ln in range(1, 17):
        for _ in range(bits[ln - 1]):
            code[vals[k]], length[vals[k]] = c, ln
            c += 1
            k += 1
        c <<= 1
    return code, length


def _write(tokens, code, length):
    acc, n, out = 0, 0, bytearray()
    for kind, v, nb in tokens:
        if kind == 0:
            v, nb = code[v], length[v]
        acc = (acc << nb) | v
        n += nb
        while n >= 8:
            b = (acc >> (n - 8)) & 0xFF
            out.append(b)
            if b == 0xFF:
                out.append(0)
            n -= 8
        acc &= (1 << n) - 1
    if n:
        b = ((acc << (8 - n)) | ((1 << (8 - n)) - 1)) & 0xFF  # pad with ones
        out.append(b)
        if b == 0xFF:
            out.append(0)
    return bytes(out)


def _category(v):
    v = abs(int(v))
    return v.bit_length()


def _dc_scan(out, planes, geom, comps, ah, al):
    """planes[c]: (bh, bw, 64) zig-zag; geom: per component (h, v); interleaved walk when more than one component."""
    H, V = geom
    if len(comps) == 1:
        c = comps[0]
        order = [(c, y, x) for y in range(planes[c].rows) for x in range(planes[c].cols)]
    else:
        mcu_h = planes[comps[0]].a.shape[0] // V[comps[0]]
        mcu_w = planes[comps[0]].a.shape[1] // H[comps[0]]
        order = [(c, my * V[c] + v, mx * H[c] + h) for my in range(mcu_h) for mx in range(mcu_w) for c in comps for v in range(V[c]) for h in range(H[c])]
    pred = {c: 0 for c in comps}
    for c, y, x in order:
        dc = int(planes[c].a[y, x, 0])
        if ah == 0:
            t = dc >> al  # (arithmetic shift, as libjpeg)
            d = t - pred[c]
            pred[c] = t
            n = _category(d)
            out.sym(n)
            out.bits(d if d >= 0 else d - 1, n)
        else:
            out.bits((dc >> al) & 1, 1)


class _Plane:
    def __init__(self, a, rows, cols):
        self.a, self.rows, self.cols = a, rows, cols


def _flush_eobrun(out, st):
    if st["eobrun"]:
        n = st["eobrun"].bit_length() - 1
        out.sym(n << 4)
        out.bits(st["eobrun"], n)
        st["eobrun"] = 0
    for b in st["be"]:
        out.bits(b, 1)
    st["be"] = []


def _ac_first(out, plane, ss, se, al):
    st = {"eobrun": 0, "be": []}
    for y in range(plane.rows):
        for x in range(plane.cols):
            blk = plane.a[y, x]
            r = 0
            for k in range(ss, se + 1):
                v = int(blk[k])
                t = (abs(v) >> al)
                if t == 0:
                    r += 1
                    continue
                _flush_eobrun(out, st)
                while r > 15:
                    out.sym(0xF0)
                    r -= 16
                n = t.bit_length()
                out.sym((r << 4) | n)
                out.bits(t if v >= 0 else ~t, n)
                r = 0
            if r > 0:
                st["eobrun"] += 1
                if st["eobrun"] == 0x7FFF:
                    _flush_eobrun(out, st)
    _flush_eobrun(out, st)


def _ac_refine(out, plane, ss, se, al):
    st = {"eobrun": 0, "be": []}
    for y in range(plane.rows):
        for x in range(plane.cols):
            blk = plane.a[y, x]
            absv = [abs(int(blk[k])) >> al for k in range(64)]
            eob = 0
            for k in range(ss, se + 1):
                if absv[k] == 1:
                    eob = k
            r, br = 0, []
            for k in range(ss, se + 1):
                t = absv[k]
                if t == 0:
                    r += 1
                    continue
                while r > 15 and k <= eob:
                    _flush_eobrun(out, st)
                    out.sym(0xF0)
                    r -= 16
                    for b in br:
                        out.bits(b, 1)
                    br = []
                if t > 1:
                    br.append(t & 1)
                    continue
                _flush_eobrun(out, st)
                out.sym((r << 4) | 1)
                out.bits(0 if int(blk[k]) < 0 else 1, 1)
                for b in br:
                    out.bits(b, 1)
                br = []
                r = 0
            if r > 0 or br:
                st["eobrun"] += 1
                st["be"] += br
                if st["eobrun"] == 0x7FFF or len(st["be"]) > 937:
                    _flush_eobrun(out, st)
    _flush_eobrun(out, st)


def encode_from_coefficients(comps, qts, coefs, width, height, script):
    """comps / qts / coefs as tools/baseline_encoder.py's encode_from_coefficients; -> progressive (SOF2) JFIF bytes."""
    ncomp = len(comps)
    H = [int(c.horizontal_sampling_factor) for c in comps]
    V = [int(c.vertical_sampling_factor) for c in comps]
    hmax, vmax = max(H), max(V)
    planes = []
    for c in range(ncomp):
        bw, bh = int(comps[c].block_width), int(comps[c].block_height)
        a = np.asarray(coefs[c], np.int64).reshape(bh, bw, 64)[:, :, UNZIGZAG]
        cw = -(-width * H[c] // hmax)
        ch = -(-height * V[c] // vmax)
        planes.append(_Plane(a, min(bh, -(-ch // 8)), min(bw, -(-cw // 8))))  # a non-interleaved scan stops at the component's edge
    if ncomp == 1:
        H, V = [1], [1]
    qts = [np.asarray(q, np.int64).reshape(64) for q in qts]
    tq = [0 if np.array_equal(q, qts[0]) else 1 for q in qts]
    ids = [1, 2, 3, 4][:ncomp]
    out = [b"\xff\xd8", _marker(0xE0, b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0")]
    for t in sorted(set(tq)):
        q = qts[tq.index(t)]
        out.append(_marker(0xDB, bytes([t]) + bytes(int(x) for x in q[UNZIGZAG])))
    out.append(_marker(0xC2, bytes([8]) + int(height).to_bytes(2, "big") + int(width).to_bytes(2, "big") + bytes([ncomp]) +
                       b"".join(bytes([ids[c], (int(comps[c].horizontal_sampling_factor) << 4) | int(comps[c].vertical_sampling_factor), tq[c]]) for c in range(ncomp))))
    for scan_comps, ss, se, ah, al in script:
        toks = _Bits()
        if ss == 0:
            _dc_scan(toks, planes, (H, V), list(scan_comps), ah, al)
        elif ah == 0:
            _ac_first(toks, planes[scan_comps[0]], ss, se, al)
        else:
            _ac_refine(toks, planes[scan_comps[0]], ss, se, al)
        syms = [t[1] for t in toks.tok if t[0] == 0]
        code, length = {}, {}
        if syms:
            freq = [0] * 256
            for s_ in syms:
                freq[s_] += 1
            bits, vals = _optimal_table(freq)
            code, length = _codes(bits, vals)
            out.append(_marker(0xC4, bytes([(0 if ss == 0 else 1) << 4]) + bytes(bits) + bytes(vals)))
        out.append(_marker(0xDA, bytes([len(scan_comps)]) + b"".join(bytes([ids[c], 0]) for c in scan_comps) + bytes([ss, se, (ah << 4) | al])))
        out.append(_write(toks.tok, code, length))
    out.append(b"\xff\xd9")
    return b"".join(out)


def encode_rgb(rgb, script, quality=85, sampling="444"):
    import jpeg_decoder_amd as J
    import synth
    h, w = rgb.shape[:2]
    samp = SAMPLINGS[sampling]
    comps, _mcu = J.make_components(w, h, samp)
    lum, chr_ = synth.quality_tables(quality)
    qts = [lum, chr_, chr_][: len(samp)]
    coefs = synth.coefficients_from_rgb(rgb if sampling != "gray" else rgb, comps, "gray" if sampling == "gray" else "ycbcr", qts)
    return encode_from_coefficients(list(comps), qts, coefs, w, h, script)


def random_script(rng, ncomp):
    """A legal script with randomly cut bands and randomly ordered refinements: per component and band a chain first -> refinements
    (Al counting down by one), DC likewise; chains are interleaved at random, each keeping its own order."""
    chains = []
    al0 = int(rng.integers(0, 3))
    dc = [(tuple(range(ncomp)) if rng.random() < 0.7 or ncomp == 1 else None, 0, 0, 0, al0)]
    if dc[0][0] is None:
        dc = [((c,), 0, 0, 0, al0) for c in range(ncomp)]
    for al in range(al0 - 1, -1, -1):
        dc.append((tuple(range(ncomp)), 0, 0, al + 1, al))
    chains.append(dc)
    for c in range(ncomp):
        cuts = sorted(set([1, 64] + [int(x) for x in rng.integers(2, 64, int(rng.integers(0, 3)))]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            al = int(rng.integers(0, 3))
            ch = [((c,), a, b - 1, 0, al)]
            for x in range(al - 1, -1, -1):
                ch.append(((c,), a, b - 1, x + 1, x))
            chains.append(ch)
    script = []
    first = chains[0].pop(0) if len(chains[0]) and chains[0][0][0] == tuple(range(ncomp)) else None
    if first:
        script.append(first)
    else:  # every component's first DC scan in front of its AC scans
        while chains[0] and chains[0][0][3] == 0:
            script.append(chains[0].pop(0))
    while any(chains):
        live = [i for i, ch in enumerate(chains) if ch]
        script.append(chains[int(rng.choice(live))].pop(0))
    return script


if __name__ == "__main__":
    import synth
    w, h = int(sys.argv[2]), int(sys.argv[3])
    data = encode_rgb(synth.synthetic_rgb(w, h), SPLIT_REFINEMENT_YCC, sampling=sys.argv[4] if len(sys.argv) > 4 else "444")
    open(sys.argv[1], "wb").write(data)
