"""CPU emulation of the product's device code against the oracle.

tests/emu compiles the SAME headers the gfx950 kernels are built from (pixel_math.hpp,
fused_core.hpp, fused_plan.hpp) with g++ and runs every workgroup / lane / phase sequentially.
This checks arithmetic, tiling, halo and edge logic here, without a GPU; the real kernels are
then checked again on the MI355X by tests/test_gpu_parity.py."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402
import oracle as O  # noqa: E402
import synth  # noqa: E402

import jpeg_decoder_amd as J  # noqa: E402  (host-side structs only; nothing here touches a GPU)


def _orc_blocks(c, q, scale=8):
    n = len(c) // 64
    return np.concatenate([O.idct_block(c[i * 64:(i + 1) * 64], q, scale).reshape(-1) for i in range(n)])


def test_device_idct_exact_path_matches_oracle():
    rng = np.random.default_rng(1)
    n = 1500
    for kind in ("full", "sparse"):
        if kind == "full":
            c = rng.integers(-32768, 32768, n * 64).astype(np.int16)
            q = rng.integers(1, 65536, 64).astype(np.uint16)
        else:
            c = synth.sparse_coefficients(rng, n)
            q = rng.integers(1, 256, 64).astype(np.uint16)
        out = np.zeros(n * 64, np.uint8)
        emu.lib().emu_idct8x8(0, c.ctypes.data, q.ctypes.data, out.ctypes.data, n)
        assert np.array_equal(out, _orc_blocks(c, q)), kind
    blocks, qts = synth.adversarial_blocks(rng)
    for c, q in zip(blocks, qts):
        out = np.zeros(64, np.uint8)
        emu.lib().emu_idct8x8(0, c.ctypes.data, q.ctypes.data, out.ctypes.data, 1)
        assert np.array_equal(out, _orc_blocks(c, q))


def test_device_idct_sane_path_exact_up_to_its_bound():
    """24-bit multiply path: exact whenever every |c*q| < 2^15 (worst case: all at the bound)."""
    rng = np.random.default_rng(2)
    n = 1500
    for trial in range(6):
        q = rng.integers(1, 256, 64).astype(np.uint16) if trial % 2 else rng.integers(1, 4096, 64).astype(np.uint16)
        lim = ((1 << 15) - 1) // q.astype(np.int64)
        sign = rng.choice([-1, 1], (n, 64))
        mag = np.broadcast_to(lim, (n, 64)) if trial < 3 else rng.integers(0, lim + 1, (n, 64))
        c = (sign * mag).astype(np.int16).reshape(-1)
        out = np.zeros(n * 64, np.uint8)
        emu.lib().emu_idct8x8(1, c.ctypes.data, q.ctypes.data, out.ctypes.data, n)
        assert np.array_equal(out, _orc_blocks(c, q)), trial


def test_device_idct_tight_path_exact_up_to_its_bound():
    """dot2 row pass: exact whenever each column's sum of |c*q| <= 5900 (column outputs then fit i16)."""
    rng = np.random.default_rng(7)
    n = 1500
    for trial in range(4):
        q = rng.integers(1, 64, 64).astype(np.uint16) if trial % 2 else np.ones(64, np.uint16)
        c = synth.tight_blocks(rng, n, q)
        out = np.zeros(n * 64, np.uint8)
        emu.lib().emu_idct8x8(2, c.ctypes.data, q.ctypes.data, out.ctypes.data, n)
        assert np.array_equal(out, _orc_blocks(c, q)), trial
    # worst case for the i16 claim: the whole budget on the coefficient with the largest gain
    for row in range(8):
        c = np.zeros((4, 8, 8), np.int16)
        c[0, row, :] = 5900
        c[1, row, :] = -5900
        c[2, row, ::2] = 5900
        c[3, row, 1::2] = -5900
        c = c.reshape(-1)
        q = np.ones(64, np.uint16)
        out = np.zeros(4 * 64, np.uint8)
        emu.lib().emu_idct8x8(2, c.ctypes.data, q.ctypes.data, out.ctypes.data, 4)
        assert np.array_equal(out, _orc_blocks(c, q)), row


@pytest.mark.parametrize("scale", [4, 2, 1])
def test_device_reduced_idct(scale):
    rng = np.random.default_rng(scale)
    n = 800
    c = rng.integers(-32768, 32768, n * 64).astype(np.int16)
    q = rng.integers(1, 65536, 64).astype(np.uint16)
    out = np.zeros(n * scale * scale, np.uint8)
    emu.lib().emu_idct_small(scale, c.ctypes.data, q.ctypes.data, out.ctypes.data, n)
    assert np.array_equal(out, _orc_blocks(c, q, scale))


def test_device_ycbcr_all_inputs():
    y, cb, cr = np.meshgrid(np.arange(256), np.arange(256), np.arange(0, 256, 3), indexing="ij")
    Y = y.astype(np.int64) * (1 << 20) + (1 << 19)
    r = np.clip((Y + 1470104 * (cr - 128)) >> 20, 0, 255)
    g = np.clip((Y - 360857 * (cb - 128) - 748830 * (cr - 128)) >> 20, 0, 255)
    b = np.clip((Y + 1858077 * (cb - 128)) >> 20, 0, 255)
    want = (r | (g << 8) | (b << 16)).astype(np.uint32)
    rng = np.random.default_rng(0)
    for _ in range(20000):
        i, j, k = rng.integers(0, 256), rng.integers(0, 256), rng.integers(0, y.shape[2])
        assert emu.lib().emu_ycbcr(int(y[i, j, k]), int(cb[i, j, k]), int(cr[i, j, k])) == int(want[i, j, k])


def test_device_ycbcr_on_centred_chroma_all_inputs():
    """The colour conversion of the fused 4:2:0 / 4:2:2 passes (chroma arrives minus 128, v_mad chains from one rounding
    term) gives src/decoder.rs:1486-1508 for all 2^24 inputs."""
    assert emu.lib().emu_ycbcr_centred_mismatches() == 0


def _to_j(ocomps):
    out = (J.Component * len(ocomps))()
    for i, c in enumerate(ocomps):
        out[i].identifier, out[i].horizontal_sampling_factor, out[i].vertical_sampling_factor = c.identifier, c.h, c.v
        out[i].quantization_table_index, out[i].dct_scale = c.tq, c.dct_scale
        out[i].size_width, out[i].size_height, out[i].block_width, out[i].block_height = c.size_w, c.size_h, c.block_w, c.block_h
    return out


def _emulate(w_, h_, samp, ct, coefs, qts, sane, seg_rows=0, s420_tx=0):
    ocomps, _ = O.make_components(w_, h_, samp)
    desc = J.image_desc(list(_to_j(ocomps)), qts, w_, h_, ct)
    n = len(samp)
    ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in coefs])
    out_len = w_ * h_ * (1 if n == 1 else n)
    out = np.full(out_len + 64, 0x5A, np.uint8)  # guard band: the kernels must not write past the image
    tx = C.c_uint32(0)
    kind = emu.lib().emu_fused_decode(C.byref(desc), ptrs, int(sane), out.ctypes.data, C.byref(tx), seg_rows, s420_tx)
    assert (out[out_len:] == 0x5A).all(), "emulated kernel wrote past the output"
    return kind, out[:out_len], tx.value


GEOMS = [
    (64, 48, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (33, 17, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (2, 2, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (3, 5, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (16, 16, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (17, 33, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (1025, 40, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (1920, 24, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (1920, 72, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (1920, 64, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (2050, 18, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (250, 130, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (672, 65, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (673, 79, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (30, 160, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (36, 20, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (38, 10, [(2, 2), (1, 1), (1, 1)], "YCbCr"),  # last column in pixel 3 / 5 of a chunk
    (13, 29, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (4, 355, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (14, 38, [(2, 2), (1, 1), (1, 1)], "YCbCr"),  # one MCU wide, several rows (tools/fuzz_gpu_geometry.py)
    (5, 40, [(1, 2), (1, 1), (1, 1)], "YCbCr"), (8, 100, [(1, 2), (1, 1), (1, 1)], "YCbCr"), (5, 40, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (7, 50, [(1, 1), (1, 1), (1, 1)], "YCbCr"),
    (36, 9, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (38, 9, [(2, 1), (1, 1), (1, 1)], "YCbCr"),
    (45, 29, [(1, 1), (1, 1), (1, 1)], "YCbCr"), (45, 29, [(1, 1), (1, 1), (1, 1)], "RGB"),
    (650, 20, [(1, 1), (1, 1), (1, 1)], "YCbCr"), (1, 1, [(1, 1), (1, 1), (1, 1)], "YCbCr"),
    (64, 24, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (33, 17, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (2, 1, [(2, 1), (1, 1), (1, 1)], "YCbCr"),
    (3, 9, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (16, 8, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (17, 8, [(2, 1), (1, 1), (1, 1)], "YCbCr"),
    (993, 10, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (1920, 16, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (2017, 9, [(2, 1), (1, 1), (1, 1)], "YCbCr"),
    (64, 48, [(1, 2), (1, 1), (1, 1)], "YCbCr"), (50, 61, [(1, 2), (1, 1), (1, 1)], "YCbCr"), (8, 2, [(1, 2), (1, 1), (1, 1)], "YCbCr"),
    (3, 5, [(1, 2), (1, 1), (1, 1)], "YCbCr"), (520, 80, [(1, 2), (1, 1), (1, 1)], "YCbCr"), (1032, 33, [(1, 2), (1, 1), (1, 1)], "YCbCr"), (17, 160, [(1, 2), (1, 1), (1, 1)], "YCbCr"),
    (64, 24, [(4, 1), (1, 1), (1, 1)], "YCbCr"), (333, 21, [(4, 1), (1, 1), (1, 1)], "YCbCr"), (3, 9, [(4, 1), (1, 1), (1, 1)], "YCbCr"), (1400, 9, [(4, 1), (1, 1), (1, 1)], "YCbCr"),
    (70, 40, [(4, 2), (1, 1), (1, 1)], "YCbCr"), (821, 17, [(4, 2), (1, 1), (1, 1)], "YCbCr"), (40, 70, [(1, 4), (1, 1), (1, 1)], "YCbCr"), (345, 33, [(1, 4), (1, 1), (1, 1)], "YCbCr"),
    (50, 61, [(2, 4), (1, 1), (1, 1)], "YCbCr"), (417, 35, [(2, 4), (1, 1), (1, 1)], "YCbCr"), (50, 61, [(4, 4), (1, 1), (1, 1)], "YCbCr"), (460, 33, [(4, 4), (1, 1), (1, 1)], "YCbCr"), (2, 2, [(4, 4), (1, 1), (1, 1)], "YCbCr"),
    (45, 29, [(1, 1)] * 4, "CMYK"), (45, 29, [(1, 1)] * 4, "YCCK"), (650, 20, [(1, 1)] * 4, "YCCK"), (513, 9, [(1, 1)] * 4, "CMYK"), (1, 1, [(1, 1)] * 4, "CMYK"),
    (37, 21, [(1, 1)], "Grayscale"), (2056, 9, [(1, 1)], "Grayscale"), (1, 1000, [(1, 1)], "Grayscale"),
    (1000, 1, [(1, 1)], "Grayscale"),
    # four components with half-size ones (fused_x4.hpp): jpg-cmyk-2.jpg's layout and YCCK with K at full size, both colour
    # functions on both; several tiles and MCU rows, widths / heights that end inside a block, an MCU, a tile; one MCU
    (65, 47, [(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK"), (65, 47, [(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK"),
    (65, 47, [(2, 2), (1, 1), (1, 1), (1, 1)], "YCCK"), (65, 47, [(2, 2), (1, 1), (1, 1), (2, 2)], "CMYK"),
    (600, 40, [(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK"), (577, 33, [(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK"),
    (16, 16, [(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK"), (2, 2, [(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK"),
    (13, 90, [(2, 2), (1, 1), (1, 1), (1, 1)], "YCCK"), (38, 10, [(2, 2), (1, 1), (1, 1), (2, 2)], "CMYK"), (290, 18, [(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK"),
    # (round 5: a strip walk like 4:2:0's — W4 — so: several strips AND several MCU rows, seams, carry rows, the closing row)
    (250, 130, [(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK"), (250, 130, [(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK"), (673, 79, [(2, 2), (1, 1), (1, 1), (1, 1)], "YCCK"),
    (1130, 50, [(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK"), (830, 66, [(2, 2), (1, 1), (1, 1), (2, 2)], "CMYK"), (30, 160, [(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK"),
    (4, 355, [(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK"), (36, 20, [(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK"),
]


@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: f"{g[0]}x{g[1]}-{len(g[2])}c{g[2][0][0]}{g[2][0][1]}-{g[3]}")
@pytest.mark.parametrize("kind", ["sane", "tight", "hostile"])
@pytest.mark.parametrize("walk_shape", ["default", "seg1", "seg3", "tx20", "tx20-seg2", "tx7", "tx7-seg1"])
def test_fused_kernel_logic_matches_oracle(geom, kind, walk_shape):
    walk = (len(geom[2]) == 3 and geom[2][0] in ((2, 2), (1, 2))) or (len(geom[2]) == 4 and geom[2][0] == (2, 2))
    if walk_shape != "default" and not walk:
        pytest.skip("the shape knobs only affect the strip walks (4:2:0 / 4:4:0)")
    # strip walks: (MCU rows per workgroup, widest strip) — short segments and narrow strips exercise seams and halos
    seg_rows, s420_tx = {"default": (1000, 0), "seg1": (1, 0), "seg3": (3, 0), "tx20": (1000, 20), "tx20-seg2": (2, 20), "tx7": (5, 7),
                         "tx7-seg1": (1, 7)}[walk_shape]
    w_, h_, samp, ct = geom
    rng = np.random.default_rng(w_ * 131 + h_)
    ocomps, _ = O.make_components(w_, h_, samp)
    if kind == "sane":
        qts = [rng.integers(1, 64, 64).astype(np.uint16) for _ in ocomps]
        coefs = [synth.sparse_coefficients(rng, c.block_w * c.block_h, amp=64, dc_amp=500) for c in ocomps]
        assert all((np.abs(c.astype(np.int64)).reshape(-1, 64) * q < (1 << 15)).all() for c, q in zip(coefs, qts))
    elif kind == "tight":
        qts = [rng.integers(1, 32, 64).astype(np.uint16) for _ in ocomps]
        coefs = [synth.tight_blocks(rng, c.block_w * c.block_h, q) for c, q in zip(ocomps, qts)]
    else:
        qts = [rng.integers(1, 65536, 64).astype(np.uint16) for _ in ocomps]
        coefs = [rng.integers(-32768, 32768, c.block_w * c.block_h * 64).astype(np.int16) for c in ocomps]
    got_kind, got, tx = _emulate(w_, h_, samp, ct, coefs, qts, {"hostile": 0, "sane": 1, "tight": 2}[kind], seg_rows, s420_tx)
    assert got_kind != 0, "planner refused a geometry the fused kernels are meant to cover"
    want = O.pixels_from_coefficients(ocomps, qts, coefs, w_, h_, ct.upper())
    assert got.size == want.size
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (f"tx={tx}", bad[:10], got[bad[:10]], want[bad[:10]])


def test_planner_keeps_odd_geometries_on_the_generic_path():
    for (w_, h_, samp, ct) in [(1, 1, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (1, 9, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
                               (1, 64, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (64, 1, [(1, 2), (1, 1), (1, 1)], "YCbCr"),
                               (64, 64, [(3, 1), (1, 1), (1, 1)], "YCbCr"), (64, 64, [(4, 1), (2, 1), (1, 1)], "YCbCr"), (64, 1, [(4, 1), (1, 1), (1, 1)], "YCbCr"), (64, 64, [(2, 1), (1, 1), (1, 1)], "RGB"), (64, 64, [(2, 2)], "Grayscale"),
                               (64, 64, [(2, 1), (1, 1), (1, 1), (1, 1)], "CMYK"), (64, 64, [(2, 2), (2, 1), (1, 1), (1, 1)], "CMYK"), (64, 64, [(1, 1)] * 4, "None")]:
        rng = np.random.default_rng(0)
        ocomps, _ = O.make_components(w_, h_, samp)
        qts = [np.ones(64, np.uint16) for _ in ocomps]
        coefs = [np.zeros(c.block_w * c.block_h * 64, np.int16) for c in ocomps]
        ocomps, _ = O.make_components(w_, h_, samp)
        desc = J.image_desc(list(_to_j(ocomps)), qts, w_, h_, ct)
        ptrs = (C.c_void_p * len(samp))(*[c.ctypes.data for c in coefs])
        out = np.zeros(w_ * h_ * 4 + 64, np.uint8)
        assert emu.lib().emu_fused_decode(C.byref(desc), ptrs, 0, out.ctypes.data, None, 0, 0) == 0


# ---- generic upsample + colour kernel (csrc/upsample_color_body.hpp + csrc/image_job.cpp) ----------------------------
UPSAMPLE_CASES = [
    (33, 17, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (1, 1, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (2, 1, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (1, 5, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (3, 3, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (16, 16, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (15, 9, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (17, 9, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (2049, 5, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (960, 24, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (961, 3, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (7, 3, [(2, 1), (1, 1), (1, 1)], "YCbCr"),
    (50, 61, [(1, 2), (1, 1), (1, 1)], "YCbCr"), (8, 2, [(1, 2), (1, 1), (1, 1)], "YCbCr"),
    (45, 29, [(1, 1), (1, 1), (1, 1)], "YCbCr"), (45, 29, [(1, 1), (1, 1), (1, 1)], "RGB"), (45, 29, [(1, 1), (1, 1), (1, 1)], "None"),
    (70, 40, [(4, 1), (1, 1), (1, 1)], "YCbCr"), (70, 41, [(4, 2), (1, 1), (2, 1)], "YCbCr"),
    (64, 48, [(1, 1), (1, 1), (1, 1), (1, 1)], "CMYK"), (65, 47, [(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK"),
    (65, 47, [(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK"), (37, 21, [(1, 1)], "Grayscale"), (40, 24, [(2, 2)], "Grayscale"),
]


@pytest.mark.parametrize("case", UPSAMPLE_CASES, ids=lambda c: f"{c[0]}x{c[1]}-{'_'.join(f'{h}{v}' for h, v in c[2])}-{c[3]}")
@pytest.mark.parametrize("scale", [8, 4, 1])
@pytest.mark.parametrize("force_slow", [0, 1], ids=["dword-path", "byte-path"])
def test_generic_upsample_color_kernel_logic_matches_oracle(case, scale, force_slow):
    w_, h_, samp, ct = case
    rng = np.random.default_rng(w_ * 7919 + h_ * 31 + scale)
    ocomps, _ = O.make_components(w_, h_, samp, dct_scale=scale)
    out_w, out_h = J.scaled_output_size(w_, h_, scale)
    planes = [rng.integers(0, 256, O.plane_bytes(c)).astype(np.uint8) for c in ocomps]
    try:
        want = O.compute_image(ocomps, planes, out_w, out_h, ct.upper())
    except O.OracleError as e:
        want = e
    jc = _to_j(ocomps)
    n = len(samp)
    ptrs = (C.c_void_p * n)(*[p.ctypes.data for p in planes])
    out_len = (ocomps[0].size_w * ocomps[0].size_h) if n == 1 else out_w * out_h * n
    out = np.full(out_len + 64, 0x5A, np.uint8)
    ln, fast = C.c_size_t(0), C.c_int(0)
    rc = emu.lib().emu_compute_image(jc, n, ptrs, out_w, out_h, J.color_transform_id(ct), out.ctypes.data, C.byref(ln), force_slow, C.byref(fast))
    if isinstance(want, O.OracleError):
        assert rc != 0
        return
    assert rc == 0 and ln.value == want.size
    assert (out[out_len:] == 0x5A).all(), "emulated kernel wrote past the output"
    assert np.array_equal(out[:out_len], want)
    if scale == 8 and not force_slow and ct not in ("None", "Grayscale") and all(max(s[0] for s in samp) // h in (1, 2) and max(s[1] for s in samp) // v in (1, 2) for h, v in samp):
        assert fast.value == 1, "planner did not take the dword path for a geometry it is meant to cover"


# ---- reduced-size decodes in one launch (csrc/fused_scaled.hpp): every tile of the launch against the oracle -------------------
SCALED_CASES = [
    (64, 48, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (33, 17, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (1, 1, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (2, 2, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (3, 5, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (17, 33, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (1930, 40, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (1025, 24, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (250, 130, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (9, 300, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (1, 40, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (40, 1, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (500, 333, [(1, 1), (1, 1), (1, 1)], "RGB"), (45, 29, [(1, 1), (1, 1), (1, 1)], "YCbCr"), (1040, 9, [(1, 1), (1, 1), (1, 1)], "YCbCr"),
    (64, 24, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (333, 21, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (3, 9, [(2, 1), (1, 1), (1, 1)], "YCbCr"),
    (50, 61, [(1, 2), (1, 1), (1, 1)], "YCbCr"), (520, 33, [(1, 2), (1, 1), (1, 1)], "YCbCr"),
    (333, 21, [(4, 1), (1, 1), (1, 1)], "YCbCr"), (70, 40, [(4, 2), (1, 1), (1, 1)], "YCbCr"), (50, 61, [(1, 4), (1, 1), (1, 1)], "YCbCr"),
    (64, 64, [(3, 1), (1, 1), (1, 1)], "YCbCr"), (70, 33, [(2, 2), (2, 1), (1, 1)], "YCbCr"),
    (45, 29, [(1, 1)] * 4, "CMYK"), (45, 29, [(1, 1)] * 4, "YCCK"), (65, 47, [(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK"), (65, 47, [(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK"),
    (64, 40, [(1, 1)] * 3, "None"), (37, 21, [(1, 1)], "Grayscale"), (2056, 9, [(1, 1)], "Grayscale"), (300, 200, [(2, 2)], "Grayscale"), (1, 1000, [(1, 1)], "Grayscale"),
]


@pytest.mark.parametrize("case", SCALED_CASES, ids=lambda g: f"{g[0]}x{g[1]}-{len(g[2])}c{g[2][0][0]}{g[2][0][1]}-{g[3]}")
@pytest.mark.parametrize("scale", [4, 2, 1])
@pytest.mark.parametrize("kind", ["sparse", "full"])
def test_scaled_fused_kernel_logic_matches_oracle(case, scale, kind):
    """fused_scaled.hpp on the CPU: tiles with their rings of neighbour blocks, every upsampler on the LDS planes through absolute
    coordinates, every colour function — against the oracle's reduced-size decode (src/idct.rs:456-565, src/upsampler.rs, Decoder::scale)."""
    w_, h_, samp, ct = case
    rng = np.random.default_rng(w_ * 17 + h_ * 3 + scale)
    ocomps, _ = O.make_components(w_, h_, samp, dct_scale=scale)
    ow, oh = J.scaled_output_size(w_, h_, scale)
    if kind == "sparse":
        qts = [rng.integers(1, 64, 64).astype(np.uint16) for _ in ocomps]
        coefs = [synth.sparse_coefficients(rng, c.block_w * c.block_h, amp=64, dc_amp=500) for c in ocomps]
    else:  # wrap-range data: the reduced IDCTs are exact in Wrapping<i32> whatever the coefficients
        qts = [rng.integers(1, 65536, 64).astype(np.uint16) for _ in ocomps]
        coefs = [rng.integers(-32768, 32768, c.block_w * c.block_h * 64).astype(np.int16) for c in ocomps]
    try:
        want = O.pixels_from_coefficients(ocomps, qts, coefs, ow, oh, ct.upper())
    except O.OracleError:
        want = None
    desc = J.image_desc(list(_to_j(ocomps)), qts, ow, oh, ct)
    n = len(samp)
    ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in coefs])
    cap = ow * oh * max(n, 1) + 64
    out = np.full(cap, 0x5A, np.uint8)
    ln, tx, tiles = C.c_size_t(0), C.c_uint32(0), C.c_uint32(0)
    rc = emu.lib().emu_scaled_fused(C.byref(desc), ptrs, out.ctypes.data, C.byref(ln), C.byref(tx), C.byref(tiles))
    if want is None:
        assert rc > 0  # the frame the reference refuses, refused the same way (build_image_job)
        return
    assert rc == 0, rc
    assert ln.value == want.size and (out[ln.value:] == 0x5A).all(), "emulated kernel wrote past the output"
    bad = np.nonzero(out[:ln.value] != want)[0]
    assert bad.size == 0, (f"tx={tx.value} tiles={tiles.value}", bad[:10], out[bad[:10]], want[bad[:10]])
    assert tx.value % 4 == 0 and tiles.value >= 1


def test_scaled_fused_planner_leaves_full_size_and_mixed_scales_alone():
    ocomps, _ = O.make_components(64, 48, [(2, 2), (1, 1), (1, 1)], dct_scale=8)
    qts = [np.ones(64, np.uint16)] * 3
    desc = J.image_desc(list(_to_j(ocomps)), qts, 64, 48, "YCbCr")
    coefs = [np.zeros(c.block_w * c.block_h * 64, np.int16) for c in ocomps]
    ptrs = (C.c_void_p * 3)(*[c.ctypes.data for c in coefs])
    out = np.zeros(64 * 48 * 3 + 64, np.uint8)
    assert emu.lib().emu_scaled_fused(C.byref(desc), ptrs, out.ctypes.data, None, None, None) == -1
