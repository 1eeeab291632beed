#!/usr/bin/env python3
"""The hand-scheduled refinement loop of the wave-per-scan progressive decoder (csrc/huff_prog_wave.hpp, pw_refine_fast) in isolation:
one wave (and one wave per SIMD) walks a synthetic block over and over; cycles per symbol at 2.4 GHz.  Needs an MI355X.
    python tools/progw_asm_bench.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_progw_asm as T  # noqa: E402 (the case structure and the entry packing)
import jpeg_decoder_amd as J  # noqa: E402

dev = C.CDLL(J._native.LIB_PATH)
dev.jpgpu_selftest_refine_fast_ms.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
dev.jpgpu_selftest_refine_fast_ms.restype = C.c_float
dev.jpgpu_selftest_refine_fast.argtypes = [C.c_void_p, C.c_uint32]


def case(nz, run):
    c = T.Case()
    c.pos, c.win, c.nx, c.dp, c.end, c.k, c.nz, c.neg, c.al, c.eob = 63, 0xAAAAAAAAAAAAAAAA & ~1, 0x55555555, 1, 64, 1, nz, nz & 0x0f0f0f0f0f0f0f0f, 0, 0
    for i in range(64):
        c.lut6[i], c.w[i], c.acc[i] = T._entry(1, 1, run, 0, 1), 0x5a5a5a5a, 0
    return c


reps = 20000
for name, nz, run in (("no corrections, run 0", 0, 0), ("every other coefficient non-zero, run 0 (a correction per symbol)", 0xAAAAAAAAAAAAAAAA, 0),
                      ("every other coefficient non-zero, run 1 (rank select + 2 corrections)", 0xAAAAAAAAAAAAAAAA, 1),
                      ("3 of 4 non-zero, run 0 (3 corrections per symbol)", 0xEEEEEEEEEEEEEEEE, 0)):
    c = case(nz, run)
    one = (T.Case * 1)()
    C.memmove(one, C.byref(c), C.sizeof(c))
    dev.jpgpu_selftest_refine_fast(C.byref(one), 1)
    syms = bin(one[0].new_nz).count("1")
    for waves in (1, 1024):
        ms = dev.jpgpu_selftest_refine_fast_ms(C.byref(c), reps, waves)
        print(f"{name}: {syms} symbols per call, {waves} wave(s): {ms:.1f} ms / {reps} calls = {ms * 1e-3 * 2.4e9 / reps / max(syms, 1):.0f} cycles per symbol")
