"""The hand-scheduled refinement loop of the wave-per-scan progressive decoder (csrc/huff_prog_wave.hpp, pw_refine_fast: inline
gfx950 assembly) against its C++ twin (the same function as tests/emu compiles it), on random states: every output — window,
bit count, position, masks, end-of-band run, exit code and the 64 accumulator lanes — must be the same.  The twin is what the CPU
tests run against the host decoder; this test is what ties the device's instructions to it."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))

pytestmark = pytest.mark.gpu


class Case(C.Structure):
    _fields_ = [("win", C.c_uint64), ("nz", C.c_uint64), ("neg", C.c_uint64), ("new_nz", C.c_uint64), ("new_neg", C.c_uint64),
                ("pos", C.c_uint32), ("nx", C.c_uint32), ("dp", C.c_uint32), ("k", C.c_uint32), ("end", C.c_uint32), ("al", C.c_uint32),
                ("eob", C.c_uint32), ("code", C.c_uint32), ("lut6", C.c_uint32 * 64), ("w", C.c_uint32 * 64), ("acc", C.c_uint32 * 64),
                ("lut8", C.c_uint16 * 256), ("table", C.c_void_p)]


def _entry(length, extra, run, kind, size=0):
    return length | (extra << 5) | (run << 10) | (kind << 17) | (size << 19)


def _random_table(rng):
    """64 entries as pw_table_load makes them for a refinement scan: new coefficients (run 0..15), end-of-band symbols, ZRL, and
    the holes (length 0: longer codes) and bad symbols the loop must hand back."""
    t = []
    for _ in range(64):
        r = rng.random()
        length = int(rng.integers(1, 7))
        if r < 0.55:
            t.append(_entry(length, 1, int(rng.integers(0, 16)) if rng.random() < 0.5 else 0, 0, 1))
        elif r < 0.75:
            e = int(rng.integers(0, 15))
            t.append(_entry(length, e, 64, 1, e))
        elif r < 0.85:
            t.append(_entry(length, 0, 15, 2))
        elif r < 0.93:
            t.append(0)
        else:
            t.append(_entry(length, 0, 0, 3, 2))
    return t


def _cases(rng, n):
    arr = (Case * n)()
    for c in arr:
        c.pos = int(rng.integers(0, 64))
        win = int(rng.integers(0, 1 << 63)) << 1 | int(rng.integers(0, 2))
        c.win = (win >> (64 - c.pos)) << (64 - c.pos) if c.pos else 0  # (nothing below the valid bits)
        c.nx = int(rng.integers(0, 1 << 32))
        c.dp = int(rng.choice([1, 2, 17, 40, 62, 63, 64, 64]))
        c.end = int(rng.choice([64, 64, 64, 6, 2, 33]))
        c.k = int(rng.integers(1, c.end)) if c.end > 1 else 1
        dens = rng.choice([0.0, 0.05, 0.3, 0.7, 0.97])
        bits = rng.random(64) < dens
        c.nz = int(sum(1 << i for i in range(1, 64) if bits[i]))
        c.neg = c.nz & int(rng.integers(0, 1 << 63))
        c.new_nz = int(rng.integers(0, 1 << 63)) & ~c.nz & ((1 << c.k) - 2)
        c.new_neg = c.new_nz & int(rng.integers(0, 1 << 63))
        c.al = int(rng.integers(0, 4))
        c.eob = int(rng.integers(0, 3))
        t = _random_table(rng)
        for i in range(256):  # the 8-bit lookup behind the holes: codes of 7 / 8 bits (new coefficient, end of band, ZRL, a bad size) or none
            r = rng.random()
            length = int(rng.integers(7, 9))
            sym = (int(rng.integers(0, 16)) << 4 | 1) if r < 0.5 else (int(rng.integers(0, 15)) << 4) if r < 0.7 else 0xF0 if r < 0.8 else (int(rng.integers(0, 16)) << 4 | int(rng.integers(2, 11))) if r < 0.87 else 0
            c.lut8[i] = (sym | length << 8) if r < 0.87 else 0
        for i in range(64):
            c.lut6[i] = t[i]
            c.w[i] = int(rng.integers(0, 1 << 32)) if rng.random() < 0.9 else 0
            c.acc[i] = int(rng.choice([0, 0, 0, 1 << c.al, (-(1 << c.al)) & 0xffffffff]))
    return arr


@pytest.mark.parametrize("seed", range(4))
def test_hand_scheduled_refinement_loop_equals_its_twin(seed):
    import emu
    import jpeg_decoder_amd as J
    assert J.device_count() >= 1
    dev = C.CDLL(J._native.LIB_PATH)
    dev.jpgpu_selftest_refine_fast.argtypes = [C.c_void_p, C.c_uint32]
    cpu = emu.lib()
    cpu.emu_progw_refine_fast.argtypes = [C.c_void_p, C.c_uint32]
    cpu.emu_progw_refine_fast.restype = None
    rng = np.random.default_rng(9100 + seed)
    n = 4096
    a = _cases(rng, n)
    b = (Case * n)()
    C.memmove(b, a, C.sizeof(a))
    cpu.emu_progw_refine_fast(C.byref(a), n)
    assert dev.jpgpu_selftest_refine_fast(C.byref(b), n) == 0
    bad = []
    codes = [0, 0, 0]
    for i in range(n):
        x, y = a[i], b[i]
        codes[x.code] += 1
        same = all(getattr(x, f) == getattr(y, f) for f in ("win", "pos", "nx", "dp", "k", "new_nz", "new_neg", "eob", "code")) and list(x.acc) == list(y.acc)
        if not same:
            bad.append(i)
    if bad:
        i = bad[0]
        x, y = a[i], b[i]
        msg = {f: (hex(getattr(x, f)), hex(getattr(y, f))) for f in ("win", "pos", "nx", "dp", "k", "new_nz", "new_neg", "eob", "code") if getattr(x, f) != getattr(y, f)}
        lanes = [(l, hex(x.acc[l]), hex(y.acc[l])) for l in range(64) if x.acc[l] != y.acc[l]]
        raise AssertionError(f"{len(bad)} of {n} states differ; case {i}: twin vs device {msg}, lanes {lanes[:8]}")
    assert min(codes) > 50, codes  # all three exits were taken


# ---- the AC first scans' loop (pw_first_fast) --------------------------------------------------------------------------------------------
class FirstCase(C.Structure):
    _fields_ = [("win", C.c_uint64), ("nz", C.c_uint64), ("neg", C.c_uint64),
                ("pos", C.c_uint32), ("nx", C.c_uint32), ("dp", C.c_uint32), ("k", C.c_uint32), ("se", C.c_uint32), ("al", C.c_uint32),
                ("eob", C.c_uint32), ("code", C.c_uint32), ("lut6", C.c_uint32 * 64), ("w", C.c_uint32 * 64), ("cf", C.c_uint32 * 64),
                ("lut8", C.c_uint16 * 256), ("table", C.c_void_p)]


def _first_cases(rng, n):
    arr = (FirstCase * n)()
    for c in arr:
        c.pos = int(rng.integers(0, 64))
        win = int(rng.integers(0, 1 << 63)) << 1 | int(rng.integers(0, 2))
        c.win = (win >> (64 - c.pos)) << (64 - c.pos) if c.pos else 0
        c.nx = int(rng.integers(0, 1 << 32))
        c.dp = int(rng.choice([1, 2, 17, 40, 62, 63, 64, 64]))
        c.se = int(rng.choice([63, 63, 63, 5, 1, 32]))
        c.k = int(rng.integers(1, c.se + 1))
        c.al = int(rng.integers(0, 5))
        c.eob = 0
        c.nz = int(rng.integers(0, 1 << 63)) & ((1 << c.k) - 2)
        c.neg = c.nz & int(rng.integers(0, 1 << 63))
        for i in range(64):
            r = rng.random()
            length = int(rng.integers(1, 7))
            if r < 0.6:     # a coefficient: size 1..12 (some too large for al: the portable path), run 0..15
                size = int(rng.integers(1, 13)) if rng.random() < 0.2 else int(rng.integers(1, 5))
                c.lut6[i] = _entry(length, size, int(rng.integers(0, 16)) if rng.random() < 0.4 else 0, 0, size)
            elif r < 0.75:  # end of band
                e = int(rng.integers(0, 15))
                c.lut6[i] = _entry(length, e, 0, 1, e)
            elif r < 0.85:  # ZRL
                c.lut6[i] = _entry(length, 0, 16, 2)
            elif r < 0.95:
                c.lut6[i] = 0
            else:
                c.lut6[i] = _entry(length, 0, 0, 3)
            c.w[i] = int(rng.integers(0, 1 << 32)) if rng.random() < 0.9 else 0
            c.cf[i] = int(rng.integers(0, 1 << 16)) if (c.nz >> i) & 1 else 0
        for i in range(256):
            r = rng.random()
            length = int(rng.integers(7, 9))
            sym = (int(rng.integers(0, 16)) << 4 | int(rng.integers(1, 11))) if r < 0.55 else (int(rng.integers(0, 15)) << 4) if r < 0.75 else 0xF0
            c.lut8[i] = (sym | length << 8) if r < 0.87 else 0
    return arr


@pytest.mark.parametrize("seed", range(4))
def test_hand_scheduled_first_scan_loop_equals_its_twin(seed):
    import emu
    import jpeg_decoder_amd as J
    assert J.device_count() >= 1
    dev = C.CDLL(J._native.LIB_PATH)
    dev.jpgpu_selftest_first_fast.argtypes = [C.c_void_p, C.c_uint32]
    cpu = emu.lib()
    cpu.emu_progw_first_fast.argtypes = [C.c_void_p, C.c_uint32]
    cpu.emu_progw_first_fast.restype = None
    rng = np.random.default_rng(9300 + seed)
    n = 4096
    a = _first_cases(rng, n)
    b = (FirstCase * n)()
    C.memmove(b, a, C.sizeof(a))
    cpu.emu_progw_first_fast(C.byref(a), n)
    assert dev.jpgpu_selftest_first_fast(C.byref(b), n) == 0
    fields = ("win", "pos", "nx", "dp", "k", "nz", "neg", "eob", "code")
    bad = [i for i in range(n) if any(getattr(a[i], f) != getattr(b[i], f) for f in fields) or list(a[i].cf) != list(b[i].cf)]
    codes = [0, 0, 0]
    for x in a:
        codes[x.code] += 1
    if bad:
        i = bad[0]
        x, y = a[i], b[i]
        msg = {f: (hex(getattr(x, f)), hex(getattr(y, f))) for f in fields if getattr(x, f) != getattr(y, f)}
        lanes = [(l, hex(x.cf[l]), hex(y.cf[l])) for l in range(64) if x.cf[l] != y.cf[l]]
        raise AssertionError(f"{len(bad)} of {n} states differ; case {i}: twin vs device {msg}, lanes {lanes[:8]}")
    assert min(codes) > 50, codes


# ---- the DC first scans' loop (pw_dc_fast) -----------------------------------------------------------------------------------------------
class DcCase(C.Structure):
    _fields_ = [("win", C.c_uint64), ("pred", C.c_uint64), ("cm0", C.c_uint64), ("cm1", C.c_uint64), ("ts0", C.c_uint64), ("ts1", C.c_uint64),
                ("pos", C.c_uint32), ("nx", C.c_uint32), ("dp", C.c_uint32), ("i", C.c_uint32), ("n", C.c_uint32), ("al", C.c_uint32),
                ("code", C.c_uint32), ("pad_", C.c_uint32), ("lut0", C.c_uint32 * 64), ("lut1", C.c_uint32 * 64), ("w", C.c_uint32 * 64), ("val", C.c_uint32 * 64)]


def _dc_cases(rng, n):
    arr = (DcCase * n)()
    for c in arr:
        c.pos = int(rng.integers(0, 64))
        win = int(rng.integers(0, 1 << 63)) << 1 | int(rng.integers(0, 2))
        c.win = (win >> (64 - c.pos)) << (64 - c.pos) if c.pos else 0
        c.nx = int(rng.integers(0, 1 << 32))
        c.dp = int(rng.choice([1, 2, 17, 40, 62, 63, 64, 64]))
        c.n = int(rng.choice([64, 64, 37, 1, 12]))
        c.i = int(rng.integers(0, c.n))
        c.al = int(rng.integers(0, 4))
        c.pred = int(rng.integers(0, 1 << 63)) << 1 | int(rng.integers(0, 2))
        ncomp = int(rng.integers(1, 5))
        comp = [int(rng.integers(0, ncomp)) for _ in range(64)]
        tab = [int(rng.choice([0, 1, 0, 1, 2])) if rng.random() < 0.1 else k & 1 for k in comp]
        c.cm0 = sum(((comp[i] >> 0) & 1) << i for i in range(c.n))
        c.cm1 = sum(((comp[i] >> 1) & 1) << i for i in range(c.n))
        c.ts0 = sum(((tab[i] >> 0) & 1) << i for i in range(c.n))
        c.ts1 = sum(((tab[i] >> 1) & 1) << i for i in range(c.n))
        for i in range(64):
            for lut in (c.lut0, c.lut1):
                r = rng.random()
                length = int(rng.integers(1, 7))
                cat = int(rng.integers(0, 12))
                lut[i] = _entry(length, cat, 0, 0, cat) if r < 0.88 else (0 if r < 0.95 else _entry(length, 0, 0, 3, 12))
            c.w[i] = int(rng.integers(0, 1 << 32)) if rng.random() < 0.9 else 0
            c.val[i] = int(rng.integers(0, 1 << 16)) if i < c.i else 0
    return arr


@pytest.mark.parametrize("seed", range(4))
def test_hand_scheduled_dc_first_loop_equals_its_twin(seed):
    import emu
    import jpeg_decoder_amd as J
    assert J.device_count() >= 1
    dev = C.CDLL(J._native.LIB_PATH)
    dev.jpgpu_selftest_dc_fast.argtypes = [C.c_void_p, C.c_uint32]
    cpu = emu.lib()
    cpu.emu_progw_dc_fast.argtypes = [C.c_void_p, C.c_uint32]
    cpu.emu_progw_dc_fast.restype = None
    rng = np.random.default_rng(9500 + seed)
    n = 4096
    a = _dc_cases(rng, n)
    b = (DcCase * n)()
    C.memmove(b, a, C.sizeof(a))
    cpu.emu_progw_dc_fast(C.byref(a), n)
    assert dev.jpgpu_selftest_dc_fast(C.byref(b), n) == 0
    fields = ("win", "pos", "nx", "dp", "i", "pred", "code")
    bad = [k for k in range(n) if any(getattr(a[k], f) != getattr(b[k], f) for f in fields) or list(a[k].val) != list(b[k].val)]
    codes = [0, 0, 0]
    for x in a:
        codes[x.code] += 1
    if bad:
        k = bad[0]
        x, y = a[k], b[k]
        msg = {f: (hex(getattr(x, f)), hex(getattr(y, f))) for f in fields if getattr(x, f) != getattr(y, f)}
        lanes = [(l, hex(x.val[l]), hex(y.val[l])) for l in range(64) if x.val[l] != y.val[l]]
        raise AssertionError(f"{len(bad)} of {n} states differ; case {k}: twin vs device {msg}, lanes {lanes[:8]}")
    assert min(codes) > 50, codes
