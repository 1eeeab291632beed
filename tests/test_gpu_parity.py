"""GPU parity tests proper: the HIP path, called through the C ABI (include/jpgpu.h), against
the CPU oracle on the same seeded inputs and against the committed golden vectors.
Bit-exact everywhere (integer / byte work)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle as O
import refimages as R
import synth

pytestmark = pytest.mark.gpu

J = None


@pytest.fixture(scope="module", autouse=True)
def _load():
    global J
    import jpeg_decoder_amd as pkg
    J = pkg
    assert J.device_count() >= 1, "no MI355X visible: the HIP path has no CPU fallback"


def to_j(comps):
    """oracle Component -> product Component (same field order and layout)."""
    out = (J.Component * len(comps))()
    for i, c in enumerate(comps):
        out[i].identifier, out[i].horizontal_sampling_factor, out[i].vertical_sampling_factor = c.identifier, c.h, c.v
        out[i].quantization_table_index, out[i].dct_scale = c.tq, c.dct_scale
        out[i].size_width, out[i].size_height, out[i].block_width, out[i].block_height = c.size_w, c.size_h, c.block_w, c.block_h
    return out


def gpu_plane(worker, comp, qt, coefs, index=0, row_by_row=False):
    per_row = comp.block_width * comp.vertical_sampling_factor * 64
    n_rows = len(coefs) // per_row
    worker.start(J.RowData(index, comp, qt))
    if row_by_row:
        for r in range(n_rows):
            worker.append_row((index, coefs[r * per_row:(r + 1) * per_row]))
    elif n_rows:
        worker.append_rows_contiguous(index, coefs, n_rows)
    return worker.get_result(index, comp)


# ---- IDCT --------------------------------------------------------------------------------
def test_idct_kats_through_worker():
    kat = json.load(open(os.path.join(R.GOLDEN, "idct_kat.json")))
    ocomps, _ = O.make_components(8, 8, [(1, 1)])
    comps = to_j(ocomps)
    with J.HipWorker() as w:
        for name in ("kat_8x8", "all_zero", "saturated"):
            k = kat[name]
            plane = gpu_plane(w, comps[0], k["quantization_table"], np.array(k["coefficients"], np.int16))
            assert list(plane) == k["expected"], name
        k = kat["h2_column_shortcut"]
        plane = gpu_plane(w, comps[0], k["quantization_table"], np.array(k["coefficients"], np.int16))
        assert list(plane[:8]) == k["expected_row0"]


def test_idct_adversarial_blocks_bit_exact():
    rng = np.random.default_rng(11)
    blocks, qts = synth.adversarial_blocks(rng)
    ocomps, _ = O.make_components(8, 8, [(1, 1)])
    comps = to_j(ocomps)
    with J.HipWorker() as w:
        for c, q in zip(blocks, qts):
            got = gpu_plane(w, comps[0], q, c)
            want = O.idct_plane(ocomps[0], q, c)
            assert np.array_equal(got, want), (c.tolist(), q.tolist())


@pytest.mark.parametrize("scale", [8, 4, 2, 1])
@pytest.mark.parametrize("kind", ["sparse", "full_range"])
def test_idct_random_planes_bit_exact(scale, kind):
    rng = np.random.default_rng(100 + scale)
    for (w_, h_, samp) in [(333, 97, [(1, 1)]), (640, 480, [(2, 2), (1, 1), (1, 1)]), (17, 1000, [(1, 2), (1, 1), (1, 1)])]:
        ocomps, _ = O.make_components(w_, h_, samp, dct_scale=scale)
        comps = to_j(ocomps)
        with J.HipWorker() as w:
            for i, oc in enumerate(ocomps):
                nblk = oc.block_w * oc.block_h
                if kind == "sparse":
                    coefs = synth.sparse_coefficients(rng, nblk)
                    qt = rng.integers(1, 256, 64).astype(np.uint16)
                else:  # uniform i16 x u16: wraps everywhere (reference: src/idct.rs:1-3)
                    coefs = rng.integers(-32768, 32768, nblk * 64).astype(np.int16)
                    qt = rng.integers(1, 65536, 64).astype(np.uint16)
                got = gpu_plane(w, comps[i], qt, coefs, index=i, row_by_row=(i == 1))
                want = O.idct_plane(oc, qt, coefs)
                assert np.array_equal(got, want), (scale, kind, w_, h_, i)


def test_partial_rows_leave_zero_plane_tail():
    """rows never appended stay 0 (results.resize(.., 0), src/worker/rayon.rs:40-49)."""
    ocomps, _ = O.make_components(64, 64, [(1, 1)])
    comps = to_j(ocomps)
    rng = np.random.default_rng(5)
    coefs = synth.sparse_coefficients(rng, 8 * 3)
    qt = rng.integers(1, 64, 64).astype(np.uint16)
    with J.HipWorker() as w:
        got = gpu_plane(w, comps[0], qt, coefs)
    want = O.idct_plane(ocomps[0], qt, coefs, n_mcu_rows=3)
    assert np.array_equal(got, want)
    assert not got[3 * 8 * 64:].any()


def test_worker_error_behaviour():
    ocomps, _ = O.make_components(16, 16, [(1, 1)])
    comps = to_j(ocomps)
    with J.HipWorker() as w:
        with pytest.raises(J.InternalError):  # append before start
            w.append_row((0, np.zeros(128, np.int16)))
        w.start(J.RowData(0, comps[0], np.ones(64, np.uint16)))
        with pytest.raises(J.InternalError):  # assert_eq!(data.len(), block_count * 64)
            w.append_row((0, np.zeros(64, np.int16)))
        assert w.get_result(3).size == 0  # mem::take of an empty Vec


# ---- upsample + colour ---------------------------------------------------------------------
CASES = [
    # (w, h, sampling, colour transform)
    (33, 17, [(2, 2), (1, 1), (1, 1)], "YCbCr"),   # H2V2 (4:2:0)
    (1, 1, [(2, 2), (1, 1), (1, 1)], "YCbCr"),     # output_width == 1 -> H1V1 override
    (2, 1, [(2, 2), (1, 1), (1, 1)], "YCbCr"),     # output_height == 1 -> v1 override (H2V1)
    (1, 5, [(2, 2), (1, 1), (1, 1)], "YCbCr"),     # H1V2 through the width-1 override
    (3, 3, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (960, 72, [(2, 1), (1, 1), (1, 1)], "YCbCr"),  # H2V1 (4:2:2)
    (50, 61, [(1, 2), (1, 1), (1, 1)], "YCbCr"),   # H1V2 (4:4:0)
    (45, 29, [(1, 1), (1, 1), (1, 1)], "YCbCr"),   # 4:4:4
    (45, 29, [(1, 1), (1, 1), (1, 1)], "RGB"),
    (45, 29, [(1, 1), (1, 1), (1, 1)], "None"),    # planar-in-row quirk (src/decoder.rs:1476-1484)
    (70, 40, [(4, 1), (1, 1), (1, 1)], "YCbCr"),   # Generic (4:1:1)
    (70, 41, [(4, 2), (1, 1), (2, 1)], "YCbCr"),   # Generic + H2V... mixes
    (64, 48, [(1, 1), (1, 1), (1, 1), (1, 1)], "CMYK"),
    (65, 47, [(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK"),   # jpg-cmyk-2 shape
    (65, 47, [(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK"),
    (37, 21, [(1, 1)], "Grayscale"),
    (40, 24, [(2, 2)], "Grayscale"),                      # stride compaction
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[1]}-{'_'.join(f'{h}{v}' for h, v in c[2])}-{c[3]}")
@pytest.mark.parametrize("scale", [8, 2])
def test_compute_image_matches_oracle(case, scale):
    w_, h_, samp, ct = case
    rng = np.random.default_rng(hash((w_, h_, len(samp), scale)) % (2 ** 32))
    ocomps, _ = O.make_components(w_, h_, samp, dct_scale=scale)
    comps = to_j(ocomps)
    out_w, out_h = J.scaled_output_size(w_, h_, scale)
    planes = [rng.integers(0, 256, O.plane_bytes(c)).astype(np.uint8) for c in ocomps]
    want = O.compute_image(ocomps, planes, out_w, out_h, ct.upper())
    got = J.compute_image_parallel(list(comps), planes, (out_w, out_h), ct)
    assert got.size == want.size
    assert np.array_equal(got, want)


def test_compute_image_errors_match_reference():
    ocomps, _ = O.make_components(16, 16, [(2, 2), (1, 1), (1, 1)])
    comps = list(to_j(ocomps))
    planes = [np.zeros(O.plane_bytes(c), np.uint8) for c in ocomps]
    for ct, exc in (("Grayscale", J.FormatError), ("CMYK", J.FormatError), ("YCCK", J.FormatError),
                    ("JcsBgYcc", J.UnsupportedError), ("Unknown", J.FormatError)):
        with pytest.raises(exc):
            J.compute_image_parallel(comps, planes, (16, 16), ct)
    with pytest.raises(J.FormatError):  # "not all components have data"
        J.compute_image_parallel(comps, [planes[0], np.zeros(0, np.uint8), planes[2]], (16, 16), "YCbCr")
    bad, _ = O.make_components(16, 16, [(3, 1), (2, 1), (1, 1)])
    with pytest.raises(J.UnsupportedError):  # NonIntegerSubsamplingRatio
        J.compute_image_parallel(list(to_j(bad)), [np.zeros(O.plane_bytes(c), np.uint8) for c in bad], (16, 16), "YCbCr")


# ---- whole path on the reference's own images ------------------------------------------------
def _gpu_pixels_from_intermediates(d, device_resident):
    comps = to_j(d.components)
    with J.HipWorker() as w:
        planes = []
        for i in range(d.ncomp):
            assert d.coefs[i] is not None
            w.start(J.RowData(i, comps[i], d.qtables[i]))
            per_row = comps[i].block_width * comps[i].vertical_sampling_factor * 64
            for r in range(len(d.coefs[i]) // per_row):
                w.append_row((i, d.coefs[i][r * per_row:(r + 1) * per_row]))
            if device_resident:
                w.finish_plane(i, i)
            else:
                planes.append(w.get_result(i, comps[i]))
        return w.compute_image(list(comps), None if device_resident else planes, (d.width, d.height), d.color_transform)


@pytest.mark.parametrize("rel", R.reftest_files(include_disabled=False))
def test_reftest_images_bit_exact_and_within_tolerance(rel):
    path = os.path.join(R.REFTEST, rel)
    d = O.decode(open(path, "rb").read(), keep_intermediates=True)
    got = _gpu_pixels_from_intermediates(d, device_resident=True)
    assert np.array_equal(got, d.pixels), rel
    key = "reftest/" + rel
    hashes = R.golden_hashes()
    if key in hashes:
        assert hashlib.sha256(got.tobytes()).hexdigest() == hashes[key]
    assert R.max_diff_vs_png(got, d.ncomp, os.path.splitext(path)[0] + ".png") <= 3


@pytest.mark.parametrize("name", ["tower.jpg", "tower_progressive.jpg", "tower_grayscale.jpg", "large_image.jpg"])
def test_bench_images_bit_exact(name):
    d = O.decode(open(os.path.join(R.GOLDEN, "benches", name), "rb").read(), keep_intermediates=True)
    got = _gpu_pixels_from_intermediates(d, device_resident=False)
    assert hashlib.sha256(got.tobytes()).hexdigest() == R.golden_hashes()["benches/" + name]


@pytest.mark.parametrize("req", [(250, 167), (125, 84), (63, 42)])
def test_scaled_decode_bit_exact(req):
    d = O.decode(open(os.path.join(R.REFTEST, "rgb.jpg"), "rb").read(), scale_to=req, keep_intermediates=True)
    got = _gpu_pixels_from_intermediates(d, device_resident=True)
    assert hashlib.sha256(got.tobytes()).hexdigest() == R.golden_hashes()[f"reftest/rgb.jpg@{req[0]}x{req[1]}"]


# ---- batch driver ------------------------------------------------------------------------------
def _batch_case(rng, w_, h_, samp, ct, kind="sparse"):
    ocomps, _ = O.make_components(w_, h_, samp)
    qts = [rng.integers(1, 200, 64).astype(np.uint16) for _ in ocomps]
    if kind == "sparse":
        coefs = [synth.sparse_coefficients(rng, c.block_w * c.block_h) for c in ocomps]
    elif kind == "tight":  # every column sum of |c*q| at the 5900 limit of the dot2 row pass
        qts = [rng.integers(1, 32, 64).astype(np.uint16) for _ in ocomps]
        coefs = [synth.tight_blocks(rng, c.block_w * c.block_h, q) for c, q in zip(ocomps, qts)]
    elif kind == "sane":  # |c*q| < 2^15 but column sums far above the tight limit
        qts = [rng.integers(1, 16, 64).astype(np.uint16) for _ in ocomps]
        coefs = [(rng.integers(-2000, 2001, c.block_w * c.block_h * 64)).astype(np.int16) for c in ocomps]
    else:
        coefs = [rng.integers(-32768, 32768, c.block_w * c.block_h * 64).astype(np.int16) for c in ocomps]
        qts = [rng.integers(1, 65536, 64).astype(np.uint16) for _ in ocomps]
    return ocomps, qts, coefs, ct, w_, h_


def _run_batch(cases, flags=0):
    descs = [J.image_desc(list(to_j(oc)), qts, w_, h_, ct) for oc, qts, _, ct, w_, h_ in cases]
    b = J.Batch(descs, flags=flags)
    for i, (oc, qts, coefs, ct, _w, _h) in enumerate(cases):
        for c in range(len(oc)):
            b.upload(i, c, coefs[c])
    b.decode()
    b.synchronize()
    outs = [b.download(i) for i in range(len(cases))]
    path = b.path
    b.close()
    return outs, path


def test_batch_heterogeneous_mixed_paths():
    rng = np.random.default_rng(77)
    cases = [
        _batch_case(rng, 64, 48, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
        _batch_case(rng, 33, 21, [(1, 1), (1, 1), (1, 1)], "RGB"),
        _batch_case(rng, 100, 9, [(1, 1)], "Grayscale"),
        _batch_case(rng, 31, 70, [(2, 1), (1, 1), (1, 1)], "YCbCr", kind="full"),
        _batch_case(rng, 16, 16, [(1, 1), (1, 1), (1, 1), (1, 1)], "CMYK"),
        _batch_case(rng, 70, 40, [(3, 1), (1, 1), (1, 1)], "YCbCr"),
        _batch_case(rng, 50, 61, [(1, 2), (1, 1), (1, 1)], "YCbCr"),
        _batch_case(rng, 70, 40, [(4, 1), (1, 1), (1, 1)], "YCbCr"),
        _batch_case(rng, 90, 50, [(4, 2), (1, 1), (1, 1)], "YCbCr"),
    ]
    outs, path = _run_batch(cases)
    assert path == "mixed"  # every fusable kind gets its own launch; 3:1:1 (and whatever else has no fused kernel) the generic kernels
    outs_g, path_g = _run_batch(cases, flags=J._native.BATCH_FORCE_GENERIC)
    assert path_g == "generic"
    outs_2, path_2 = _run_batch(cases[4:])
    assert path_2 == "mixed"
    assert _run_batch(cases[5:6])[1] == "generic"
    for (oc, qts, coefs, ct, w_, h_), got, gen in zip(cases, outs, outs_g):
        want = O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct.upper())
        assert np.array_equal(got, want)
        assert np.array_equal(gen, want)
    for (oc, qts, coefs, ct, w_, h_), got in zip(cases[4:], outs_2):
        assert np.array_equal(got, O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct.upper()))


SAME_GEOMETRY = [
    (64, 48, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (33, 17, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (1, 1, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (250, 130, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (129, 257, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (13, 29, [(2, 2), (1, 1), (1, 1)], "YCbCr"),  # one MCU wide, several segments: the seam round needs more staging than a step
    (4, 355, [(2, 2), (1, 1), (1, 1)], "YCbCr"),
    (5, 100, [(1, 2), (1, 1), (1, 1)], "YCbCr"),
    (45, 29, [(1, 1), (1, 1), (1, 1)], "YCbCr"),
    (200, 120, [(1, 1), (1, 1), (1, 1)], "YCbCr"),
    (64, 24, [(2, 1), (1, 1), (1, 1)], "YCbCr"),
    (993, 21, [(2, 1), (1, 1), (1, 1)], "YCbCr"),
    (2, 9, [(2, 1), (1, 1), (1, 1)], "YCbCr"),
    (37, 21, [(1, 1)], "Grayscale"),
    (300, 200, [(1, 1)], "Grayscale"),
    (333, 45, [(4, 1), (1, 1), (1, 1)], "YCbCr"),  # UpsamplerGeneric layouts: fusedgen
    (150, 70, [(4, 2), (1, 1), (1, 1)], "YCbCr"),
    (61, 150, [(1, 4), (1, 1), (1, 1)], "YCbCr"),
    (130, 97, [(2, 4), (1, 1), (1, 1)], "YCbCr"),
    (200, 129, [(4, 4), (1, 1), (1, 1)], "YCbCr"),
]


@pytest.mark.parametrize("case", SAME_GEOMETRY, ids=lambda c: f"{c[0]}x{c[1]}-{len(c[2])}c-{c[2][0][0]}{c[2][0][1]}")
@pytest.mark.parametrize("kind", ["sparse", "tight", "sane", "full"])
def test_batch_same_geometry_fast_path_bit_exact(case, kind):
    """Same-geometry batches resolve to the fused kernels; results must equal the oracle and the
    generic path byte for byte, for sane (24-bit multiply path) and hostile (wrap-exact) data."""
    w_, h_, samp, ct = case
    rng = np.random.default_rng(w_ * 1000 + h_)
    cases = [_batch_case(rng, w_, h_, samp, ct, kind=kind) for _ in range(5)]
    # all images of a batch share q-tables only by accident; keep them distinct on purpose
    outs, path = _run_batch(cases)
    outs_generic, path_g = _run_batch(cases, flags=J._native.BATCH_FORCE_GENERIC)
    assert path_g == "generic"
    for (oc, qts, coefs, ct_, _w, _h), got, gen in zip(cases, outs, outs_generic):
        want = O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper())
        assert np.array_equal(gen, want), "generic path"
        assert np.array_equal(got, want), f"path {path}"


def test_batch_1080p_420_full_size_properties():
    """BASELINE configs[1] geometry at full size: oracle on 2 images, then size-independent
    properties over the whole batch (identical inputs -> identical outputs; distinct seeds differ)."""
    rng = np.random.default_rng(2024)
    w_, h_ = 1920, 1080
    ocomps, _ = O.make_components(w_, h_, [(2, 2), (1, 1), (1, 1)])
    jc = to_j(ocomps)
    lum, chr_ = synth.quality_tables(85)
    qts = [lum, chr_, chr_]
    rgb = synth.synthetic_rgb(w_, h_)
    base = synth.coefficients_from_rgb(rgb, jc, "ycbcr", qts)
    other = [synth.sparse_coefficients(rng, c.block_w * c.block_h) for c in ocomps]
    n = 12
    cases = [(ocomps, qts, other if i == 5 else base, "YCbCr", w_, h_) for i in range(n)]
    outs, path = _run_batch(cases)
    want_base = O.pixels_from_coefficients(ocomps, qts, base, w_, h_, "YCBCR")
    want_other = O.pixels_from_coefficients(ocomps, qts, other, w_, h_, "YCBCR")
    digest = hashlib.sha256(want_base.tobytes()).hexdigest()
    for i, got in enumerate(outs):
        if i == 5:
            assert np.array_equal(got, want_other)
        else:
            assert hashlib.sha256(got.tobytes()).hexdigest() == digest, (i, path)
    # the decoded synthetic image is close to its source (sanity of the whole chain)
    err = np.abs(outs[0].reshape(h_, w_, 3).astype(int) - rgb.astype(int))
    assert err.mean() < 6.0


def test_batch_2160p_420_full_size_properties():
    """BASELINE configs[2] geometry (3840x2160 4:2:0, the per-GPU shard of the 4096-image job) at full size: the
    oracle on two distinct images, then identical inputs -> identical outputs over the rest of the batch."""
    rng = np.random.default_rng(2160)
    w_, h_ = 3840, 2160
    ocomps, _ = O.make_components(w_, h_, [(2, 2), (1, 1), (1, 1)])
    jc = to_j(ocomps)
    lum, chr_ = synth.quality_tables(85)
    qts = [lum, chr_, chr_]
    rgb = synth.synthetic_rgb(w_, h_)
    base = synth.coefficients_from_rgb(rgb, jc, "ycbcr", qts)
    other = [synth.sparse_coefficients(rng, c.block_w * c.block_h) for c in ocomps]
    n = 6
    cases = [(ocomps, qts, other if i == 3 else base, "YCbCr", w_, h_) for i in range(n)]
    outs, path = _run_batch(cases)
    assert path.startswith("fused420"), path
    want_base = O.pixels_from_coefficients(ocomps, qts, base, w_, h_, "YCBCR")
    want_other = O.pixels_from_coefficients(ocomps, qts, other, w_, h_, "YCBCR")
    digest = hashlib.sha256(want_base.tobytes()).hexdigest()
    for i, got in enumerate(outs):
        if i == 3:
            assert np.array_equal(got, want_other)
        else:
            assert hashlib.sha256(got.tobytes()).hexdigest() == digest, (i, path)
    err = np.abs(outs[0].reshape(h_, w_, 3).astype(int) - rgb.astype(int))
    assert err.mean() < 6.0


STRIP_GEOMETRY = [(64, 48), (33, 17), (2, 2), (250, 130), (129, 257), (673, 79), (1920, 64), (30, 160)]


@pytest.mark.parametrize("size", STRIP_GEOMETRY, ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("knobs", [{}, {"JPGPU_S420_SEG": "1"}, {"JPGPU_S420_SEG": "3", "JPGPU_S420_TX": "20"}, {"JPGPU_S420_TX": "7"},
                                   {"JPGPU_S420_CUTS": "3,4,9"}, {"JPGPU_S420_CUTS": "1,2,5,6,12,13", "JPGPU_S420_TX": "11"}],
                         ids=["default", "seg1", "seg3-tx20", "tx7", "unequal-segments-longest-first", "unequal-segments-tx11"])
@pytest.mark.parametrize("kind", ["sparse", "tight", "full"])
def test_batch_420_strip_walk_variant_bit_exact(size, knobs, kind, monkeypatch):
    """The 4:2:0 strip walk: strip / segment seams, carry rows, image edges."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    w_, h_ = size
    rng = np.random.default_rng(w_ * 77 + h_)
    cases = [_batch_case(rng, w_, h_, [(2, 2), (1, 1), (1, 1)], "YCbCr", kind=kind) for _ in range(3)]
    outs, path = _run_batch(cases)
    assert path == "fused420"
    for (oc, qts, coefs, ct_, _w, _h), got in zip(cases, outs):
        want = O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper())
        assert np.array_equal(got, want)


MIXED_SIZES = {
    "420": ([(2, 2), (1, 1), (1, 1)], "YCbCr", "fused420", [(64, 48), (33, 17), (2, 2), (250, 130), (129, 257), (1920, 64), (17, 1080), (640, 480), (36, 20), (38, 20), (3, 5), (30, 40)]),
    "444": ([(1, 1), (1, 1), (1, 1)], "YCbCr", "fused444", [(45, 29), (200, 120), (1, 1), (513, 8), (9, 300), (640, 480)]),
    "444rgb": ([(1, 1), (1, 1), (1, 1)], "RGB", "fused444", [(45, 29), (8, 8), (700, 33)]),
    "422": ([(2, 1), (1, 1), (1, 1)], "YCbCr", "fused422", [(64, 24), (33, 17), (2, 1), (1000, 9), (17, 300), (1984, 8), (1985, 8), (640, 480), (36, 9), (38, 9), (3, 9), (30, 17)]),
    "gray": ([(1, 1)], "Grayscale", "fusedgray", [(37, 21), (300, 200), (1, 1000), (2056, 9), (8, 8)]),
    "440": ([(1, 2), (1, 1), (1, 1)], "YCbCr", "fused440", [(64, 48), (50, 61), (8, 2), (3, 5), (520, 80), (17, 1080), (640, 480), (1032, 33)]),
    "cmyk": ([(1, 1)] * 4, "CMYK", "fused444x4", [(45, 29), (200, 120), (1, 1), (513, 8), (9, 300), (640, 480)]),
    "ycck": ([(1, 1)] * 4, "YCCK", "fused444x4", [(45, 29), (8, 8), (700, 33), (500, 333)]),
}


@pytest.mark.parametrize("key", sorted(MIXED_SIZES))
@pytest.mark.parametrize("kind", ["sparse", "tight", "full"])
def test_batch_mixed_sizes_of_one_kind_take_the_fused_path(key, kind):
    """Images of one fusable kind but different sizes: per-image geometry + work tables, same kernels, bit-exact."""
    samp, ct, want_path, sizes = MIXED_SIZES[key]
    rng = np.random.default_rng(len(key) * 1009 + len(kind))
    cases = [_batch_case(rng, w_, h_, samp, ct, kind=kind) for (w_, h_) in sizes]
    outs, path = _run_batch(cases)
    assert path == want_path
    for (oc, qts, coefs, ct_, w_, h_), got in zip(cases, outs):
        want = O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper())
        assert np.array_equal(got, want), (w_, h_)


@pytest.mark.parametrize("case", [(250, 130, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (200, 120, [(1, 1), (1, 1), (1, 1)], "YCbCr"),
                                  (300, 200, [(1, 1)], "Grayscale")], ids=["420", "444", "gray"])
def test_uniform_batch_through_the_work_table_form(case, monkeypatch):
    """Uniform batches normally use the 3-D grid; JPGPU_FUSED_TABLE=1 sends them through the work tables as well."""
    monkeypatch.setenv("JPGPU_FUSED_TABLE", "1")
    w_, h_, samp, ct = case
    rng = np.random.default_rng(w_ + h_)
    cases = [_batch_case(rng, w_, h_, samp, ct) for _ in range(4)]
    outs, path = _run_batch(cases)
    assert path.startswith("fused")
    for (oc, qts, coefs, ct_, _w, _h), got in zip(cases, outs):
        assert np.array_equal(got, O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper()))


def test_batch_of_different_kinds_runs_one_fused_launch_per_kind():
    rng = np.random.default_rng(4)
    cases = [_batch_case(rng, 64, 48, [(2, 2), (1, 1), (1, 1)], "YCbCr"), _batch_case(rng, 64, 48, [(1, 1), (1, 1), (1, 1)], "YCbCr"),
             _batch_case(rng, 64, 48, [(1, 1), (1, 1), (1, 1)], "RGB"), _batch_case(rng, 30, 20, [(1, 1)], "Grayscale")]
    outs, path = _run_batch(cases)
    assert path == "mixed"
    for (oc, qts, coefs, ct_, w_, h_), got in zip(cases, outs):
        assert np.array_equal(got, O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper()))
    outs, path = _run_batch(cases[1:3])  # 4:4:4 YCbCr next to 4:4:4 RGB: different colour functions, two launches
    assert path == "mixed"
    for (oc, qts, coefs, ct_, w_, h_), got in zip(cases[1:3], outs):
        assert np.array_equal(got, O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper()))


@pytest.mark.parametrize("case", [(250, 130, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (200, 120, [(1, 1), (1, 1), (1, 1)], "YCbCr"),
                                  (64, 24, [(2, 1), (1, 1), (1, 1)], "YCbCr"), (300, 200, [(1, 1)], "Grayscale"),
                                  (65, 47, [(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK"), (70, 40, [(3, 1), (1, 1), (1, 1)], "YCbCr")],
                         ids=["420", "444", "422", "gray", "ycck-x4", "311-generic"])
@pytest.mark.parametrize("kind", ["sparse", "full"])
def test_batch_compact_upload_equals_dense_upload(case, kind):
    """Compact coefficient transport (bitmap + index + values over PCIe, expand kernel on the device): same pixels as
    the dense upload, for sparse JPEG-like data and for blocks without a single zero."""
    w_, h_, samp, ct = case
    rng = np.random.default_rng(w_ * 3 + h_)
    cases = [_batch_case(rng, w_, h_, samp, ct, kind=kind) for _ in range(3)]
    descs = [J.image_desc(list(to_j(oc)), qts, w_, h_, ct) for oc, qts, _, ct, w_, h_ in cases]
    b = J.Batch(descs)
    sent = dense = 0
    for i, (oc, qts, coefs, _ct, _w, _h) in enumerate(cases):
        for c in range(len(oc)):
            sent += b.upload_compact(i, c, coefs[c])
            dense += coefs[c].size * 2
    b.decode()
    b.synchronize()
    for i, (oc, qts, coefs, ct_, _w, _h) in enumerate(cases):
        assert np.array_equal(b.download(i), O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper()))
    if kind == "sparse":
        assert sent < dense / 2
    # a second decode without new uploads works on the expanded arena; dense uploads afterwards still work
    b.decode()
    b.synchronize()
    oc, qts, coefs, ct_, _w, _h = cases[0]
    assert np.array_equal(b.download(0), O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper()))
    oc1, qts1, coefs1, ct1, _w, _h = cases[1]
    for c in range(len(oc1)):
        b.upload(1, c, coefs1[c][::-1].copy())  # something else, densely
    b.decode()
    b.synchronize()
    assert np.array_equal(b.download(1), O.pixels_from_coefficients(oc1, qts1, [x[::-1].copy() for x in coefs1], w_, h_, ct1.upper()))
    assert np.array_equal(b.download(0), O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper()))
    b.close()


def test_batch_compact_upload_rejects_inconsistent_buffers():
    rng = np.random.default_rng(1)
    oc, qts, coefs, ct, w_, h_ = _batch_case(rng, 64, 48, [(1, 1)], "Grayscale")
    b = J.Batch([J.image_desc(list(to_j(oc)), qts, w_, h_, ct)])
    L = J.lib()
    nblk = coefs[0].size // 64
    buf = np.zeros(L.jpgpu_compact_max_bytes(nblk), np.uint8)
    n = L.jpgpu_compact_encode(coefs[0].ctypes.data, nblk, qts[0].ctypes.data, buf.ctypes.data, None)
    assert L.jpgpu_batch_upload_compact(b._h, 0, 0, buf.ctypes.data, n - 2, -1, None) == J._native.ERR_FORMAT   # short
    bad = buf.copy()
    bad[8 * nblk + 4] ^= 1  # index of block 1
    assert L.jpgpu_batch_upload_compact(b._h, 0, 0, bad.ctypes.data, n, -1, None) == J._native.ERR_FORMAT
    assert L.jpgpu_batch_upload_compact(b._h, 0, 0, buf.ctypes.data, 12 * nblk - 1, -1, None) == J._native.ERR_FORMAT
    assert L.jpgpu_batch_upload_compact(b._h, 0, 0, buf.ctypes.data, n, -1, None) == 0
    b.synchronize()
    b.close()


@pytest.mark.parametrize("name,samp,mode,ct,path", [
    ("422", [(2, 1), (1, 1), (1, 1)], "ycbcr", "YCbCr", "fused422"), ("444", [(1, 1), (1, 1), (1, 1)], "ycbcr", "YCbCr", "fused444"),
    ("gray", [(1, 1)], "gray", "Grayscale", "fusedgray"), ("cmyk", [(1, 1)] * 4, "cmyk", "CMYK", "fused444x4"),
    ("440", [(1, 2), (1, 1), (1, 1)], "ycbcr", "YCbCr", "fused440")])
def test_batch_1080p_other_kinds_full_size(name, samp, mode, ct, path):
    """BASELINE geometry at full size for the other fused kinds: oracle on the decoded synthetic image, identical
    inputs -> identical outputs across the batch, and the decode is close to its source."""
    w_, h_ = 1920, 1080
    ocomps, _ = O.make_components(w_, h_, samp)
    jc = to_j(ocomps)
    lum, chr_ = synth.quality_tables(85)
    qts = [lum] * 4 if mode == "cmyk" else [lum, chr_, chr_][: len(samp)]
    rgb = synth.synthetic_rgb(w_, h_)
    base = synth.coefficients_from_rgb(rgb, jc, mode, qts)
    cases = [(ocomps, qts, base, ct, w_, h_)] * 5
    outs, got_path = _run_batch(cases)
    assert got_path == path
    want = O.pixels_from_coefficients(ocomps, qts, base, w_, h_, ct.upper())
    for got in outs:
        assert np.array_equal(got, want)
    if mode == "gray":
        y = 0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2]
        err = np.abs(outs[0].reshape(h_, w_).astype(float) - y)
    elif mode == "cmyk":  # the planes were the primaries themselves: 255 - decoded = source
        err = np.abs(255 - outs[0].reshape(h_, w_, 4)[..., :3].astype(int) - rgb.astype(int))
    else:
        err = np.abs(outs[0].reshape(h_, w_, 3).astype(int) - rgb.astype(int))
    assert err.mean() < 6.0


def test_batch_1080p_444_and_gray_interleaved_full_size():
    """BASELINE configs[4] at full size: 1920x1080 4:4:4 and grayscale images interleaved in ONE batch (two fused launch groups,
    path "mixed"), every image equal to the oracle, whichever group it sits in and wherever in the batch."""
    w_, h_ = 1920, 1080
    lum, chr_ = synth.quality_tables(85)
    rgb = synth.synthetic_rgb(w_, h_)
    oc444, _ = O.make_components(w_, h_, [(1, 1)] * 3)
    ocg, _ = O.make_components(w_, h_, [(1, 1)])
    c444 = synth.coefficients_from_rgb(rgb, to_j(oc444), "ycbcr", [lum, chr_, chr_])
    cg = synth.coefficients_from_rgb(rgb, to_j(ocg), "gray", [lum])
    cases = [(oc444, [lum, chr_, chr_], c444, "YCbCr", w_, h_) if i % 2 == 0 else (ocg, [lum], cg, "Grayscale", w_, h_) for i in range(12)]
    outs, path = _run_batch(cases)
    assert path == "mixed"
    want444 = O.pixels_from_coefficients(oc444, [lum, chr_, chr_], c444, w_, h_, "YCBCR")
    wantg = O.pixels_from_coefficients(ocg, [lum], cg, w_, h_, "GRAYSCALE")
    for i, got in enumerate(outs):
        assert np.array_equal(got, want444 if i % 2 == 0 else wantg), i


def test_batch_scan_ranges_on_device_equals_host_classification():
    """jpgpu_batch_scan_ranges (the pass the device entropy decoder relies on: the host never sees those coefficients)
    against jpgpu_range_class on the same planes: every class, the exact boundaries (|c*q| = 2^15 - 1 / 2^15, column sum
    5900 / 5901), the extreme value in the first / last / a middle block of planes whose block counts are not multiples of
    the kernel's tile, 16-bit quantization tables."""
    import ctypes as C
    from jpeg_decoder_amd import _native as N
    rng = np.random.default_rng(77)
    geoms = [(250, 130, [(2, 2), (1, 1), (1, 1)], "YCbCr"), (8, 8, [(1, 1)], "Grayscale"), (1000, 600, [(1, 1), (1, 1), (1, 1)], "YCbCr"),
             (300, 200, [(1, 1)], "Grayscale")]
    cases = []
    for gi, (w_, h_, samp, ct) in enumerate(geoms):
        for variant in range(6):
            oc, qts, coefs, ct_, _w, _h = _batch_case(rng, w_, h_, samp, ct, kind="sparse")
            coefs = [np.clip(c, -20, 20) for c in coefs]  # class 3 to start with (q <= 255 in _batch_case? keep sums small)
            qts = [np.minimum(q, 16).astype(np.uint16) for q in qts]
            for c in range(len(coefs)):
                plane = coefs[c].reshape(-1, 64)
                plane[:] = np.where(rng.random(plane.shape) < 0.1, plane, 0)
                nb = plane.shape[0]
                where = [0, nb - 1, nb // 2, (nb * 2) // 3, 0, nb - 1][variant] if nb > 1 else 0
                q = qts[c].reshape(64).astype(np.int64)
                if variant == 1:    # one column sum of exactly 5900 -> class 3 still
                    plane[where, :] = 0
                    q[3::8] = 1
                    plane[where, 3::8] = [737, 737, 737, 737, 738, 738, 738, 738]
                elif variant == 2:  # 5901 -> class 1
                    plane[where, :] = 0
                    q[3::8] = 1
                    plane[where, 3::8] = [737, 737, 737, 738, 738, 738, 738, 738]
                elif variant == 3:  # |c*q| = 32767 -> class 1
                    plane[where, :] = 0
                    q[63] = 1
                    plane[where, 63] = -32767
                elif variant == 4:  # |c*q| = 32768 -> class 0
                    plane[where, :] = 0
                    q[9] = 2
                    plane[where, 9] = 16384
                elif variant == 5:  # 16-bit table entry times a small coefficient -> class 0
                    plane[where, :] = 0
                    q[17] = 65535
                    plane[where, 17] = -1
                qts[c] = q.astype(np.uint16).reshape(qts[c].shape)
                coefs[c] = plane.reshape(coefs[c].shape).astype(np.int16)
            cases.append((oc, qts, coefs, ct_, w_, h_))
    descs = [J.image_desc(list(to_j(oc)), qts, w_, h_, ct) for oc, qts, _, ct, w_, h_ in cases]
    b = J.Batch(descs)
    want = np.zeros((len(cases), 4), np.uint8)
    for i, (oc, qts, coefs, _ct, _w, _h) in enumerate(cases):
        for c in range(len(oc)):
            b.upload(i, c, coefs[c])
            a = np.ascontiguousarray(coefs[c], np.int16).reshape(-1)
            q = np.ascontiguousarray(qts[c], np.uint16).reshape(64)
            want[i, c] = N.lib().jpgpu_range_class(a.ctypes.data, a.size, q.ctypes.data)
            b._check(N.lib().jpgpu_batch_set_range_class(b._h, i, c, 0))  # forget what upload() found out
    assert set(np.unique(want)) == {0, 1, 3}
    got = b.scan_ranges()
    assert np.array_equal(got, want), np.argwhere(got != want)
    b.decode()
    b.synchronize()
    for i, (oc, qts, coefs, ct_, w_, h_) in enumerate(cases):
        assert np.array_equal(b.download(i), O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper())), i
    b.close()


@pytest.mark.parametrize("samp,ct", [([(2, 2), (1, 1), (1, 1)], "YCbCr"), ([(1, 1), (1, 1), (1, 1)], "YCbCr"), ([(2, 1), (1, 1), (1, 1)], "YCbCr"),
                                     ([(1, 1)], "Grayscale"), ([(1, 2), (1, 1), (1, 1)], "YCbCr"), ([(1, 1)] * 4, "YCCK")],
                         ids=["420", "444", "422", "gray", "440", "ycck"])
def test_one_hostile_image_costs_only_itself(samp, ct):
    """A launch group whose images disagree on the arithmetic class is split per class: one image with
    wrap-range coefficients and one of class 1 among 64 leave the other 62 on the class-3 kernels, and every image —
    whichever kernel took it — equals the oracle."""
    rng = np.random.default_rng(len(samp) * 1000 + samp[0][0] * 10 + samp[0][1])
    w_, h_ = 200, 120
    cases = [_batch_case(rng, w_, h_, samp, ct, kind="tight") for _ in range(64)]
    cases[17] = _batch_case(rng, w_, h_, samp, ct, kind="full")
    cases[40] = _batch_case(rng, w_, h_, samp, ct, kind="sane")
    descs = [J.image_desc(list(to_j(oc)), qts, w, h, ct_) for oc, qts, _c, ct_, w, h in cases]
    b = J.Batch(descs)
    try:
        for i, (oc, qts, coefs, ct_, _w, _h) in enumerate(cases):
            for c in range(len(coefs)):
                b.upload(i, c, coefs[c])
        exact, sane, tight = b.class_counts()
        assert exact == 1 and tight >= 62 and exact + sane + tight == 64, (exact, sane, tight)
        b.decode()
        b.synchronize()
        assert b.path.startswith("fused")
        for i, (oc, qts, coefs, ct_, _w, _h) in enumerate(cases):
            assert np.array_equal(b.download(i), O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper())), i
        # the device-side scan arrives at the same split
        b.scan_ranges()
        assert b.class_counts() == (exact, sane, tight)
    finally:
        b.close()


@pytest.mark.parametrize("case,path", [((250, 130, [(2, 2), (1, 1), (1, 1)], "YCbCr"), "fused420"), ((200, 120, [(1, 1)] * 3, "YCbCr"), "fused444"),
                                       ((64, 24, [(2, 1), (1, 1), (1, 1)], "YCbCr"), "fused422"), ((50, 61, [(1, 2), (1, 1), (1, 1)], "YCbCr"), "fused440"),
                                       ((37, 21, [(1, 1)], "Grayscale"), "fusedgray"), ((45, 29, [(1, 1)] * 4, "CMYK"), "fused444x4"),
                                       ((70, 40, [(4, 1), (1, 1), (1, 1)], "YCbCr"), "fusedgen"),
                                       ((70, 40, [(3, 1), (1, 1), (1, 1)], "YCbCr"), "generic"),
                                       ((65, 47, [(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK"), "fused420x4-2211"),
                                       ((65, 47, [(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK"), "fused420x4-2212")],
                         ids=["420", "444", "422", "440", "gray", "cmyk", "411", "311", "cmyk-2211", "ycck-2212"])
@pytest.mark.parametrize("kind", ["sparse", "full"])
def test_worker_device_resident_flow_takes_the_fused_kernels(case, path, kind):
    """The drop-in surface (start / append_row / finish_plane / compute_image, what rust/src/worker/hip.rs calls): complete
    planes at full scale stay coefficients until compute_image, which runs the fused kernel of the frame's kind on them
    (VERDICT r1 weak #7); a plane that is short of rows, or wanted back by get_result, is transformed the generic way."""
    w_, h_, samp, ct = case
    rng = np.random.default_rng(w_ * 7 + h_)
    oc, qts, coefs, _ct, _w, _h = _batch_case(rng, w_, h_, samp, ct, kind=kind)
    comps = to_j(oc)
    want = O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct.upper())
    with J.HipWorker() as w:
        for rep in range(2):  # the second image of the same geometry reuses the worker's plan
            for i in range(len(samp)):
                w.start(J.RowData(i, comps[i], qts[i]))
                per_row = comps[i].block_width * comps[i].vertical_sampling_factor * 64
                for r in range(len(coefs[i]) // per_row):
                    w.append_row((i, coefs[i][r * per_row:(r + 1) * per_row]))
                w.finish_plane(i, i)
            got = w.compute_image(list(comps), None, (w_, h_), ct)
            assert w.last_path == path
            assert np.array_equal(got, want)
        # a short plane (a scan that ended early) cannot take that route: zero tail, generic kernels
        for i in range(len(samp)):
            w.start(J.RowData(i, comps[i], qts[i]))
            per_row = comps[i].block_width * comps[i].vertical_sampling_factor * 64
            rows = len(coefs[i]) // per_row
            for r in range(rows - 1 if i == 0 and rows > 1 else rows):
                w.append_row((i, coefs[i][r * per_row:(r + 1) * per_row]))
            w.finish_plane(i, i)
        got = w.compute_image(list(comps), None, (w_, h_), ct)
        rows0 = len(coefs[0]) // (comps[0].block_width * comps[0].vertical_sampling_factor * 64)
        if rows0 > 1:
            assert w.last_path == "generic"
            planes = [O.idct_plane(oc[i], qts[i], coefs[i], n_mcu_rows=(rows0 - 1 if i == 0 else None)) for i in range(len(samp))]
            assert np.array_equal(got, O.compute_image(oc, planes, w_, h_, ct.upper()))


def test_batch_quantization_table_replaced_after_upload():
    """jpgpu_batch_set_quantization_table after the coefficients went up: the class computed at upload time (with the old,
    small table: tight) must not survive a table under which the products leave the 16-bit range — the decode has to equal
    the oracle's with the NEW table (wrap-exact arithmetic), for the strip walk and for a tile kernel."""
    rng = np.random.default_rng(404)
    for samp in ([(2, 2), (1, 1), (1, 1)], [(1, 1), (1, 1), (1, 1)]):
        w_, h_ = 80, 48
        ocomps, _ = O.make_components(w_, h_, samp)
        small = [np.full(64, 1, np.uint16) for _ in ocomps]
        big = [rng.integers(2000, 65536, 64).astype(np.uint16) for _ in ocomps]
        coefs = [rng.integers(-300, 301, c.block_w * c.block_h * 64).astype(np.int16) for c in ocomps]
        desc = J.image_desc(list(to_j(ocomps)), small, w_, h_, "YCbCr")
        b = J.Batch([desc, desc])
        for i in range(2):
            for c in range(3):
                b.upload(i, c, coefs[c])
        assert b.class_counts()[2] == 2  # both images tight under the table of ones
        for c in range(3):
            b.set_quantization_table(1, c, big[c])  # image 1 only
        b.decode()
        b.synchronize()
        got0, got1 = b.download(0), b.download(1)
        assert b.class_counts()[0] == 1 and b.class_counts()[2] == 1
        b.close()
        assert np.array_equal(got0, O.pixels_from_coefficients(ocomps, small, coefs, w_, h_, "YCBCR"))
        assert np.array_equal(got1, O.pixels_from_coefficients(ocomps, big, coefs, w_, h_, "YCBCR"))


# ---- classes decided ON THE DEVICE (csrc/range_stats.hpp): no host between the writer of the coefficients and the pixel kernels ----
def _exact_image_class(qts, coefs):
    import ctypes as C  # noqa: F401
    from jpeg_decoder_amd import _native as N
    cls = 3
    for q, c in zip(qts, coefs):
        a = np.ascontiguousarray(c, np.int16).reshape(-1)
        qq = np.ascontiguousarray(q, np.uint16).reshape(64)
        cls = min(cls, N.lib().jpgpu_range_class(a.ctypes.data, a.size, qq.ctypes.data))
    return cls


def _by_product_image_class(qts, coefs):
    """Class drawn from max |DC*q| and max |AC*q| alone (what writers that see one coefficient at a time leave behind)."""
    max_dc = max_ac = 0
    for q, c in zip(qts, coefs):
        s = np.abs(np.asarray(c, np.int64).reshape(-1, 64) * np.asarray(q, np.int64).reshape(64))
        max_dc = max(max_dc, int(s[:, 0].max()))
        max_ac = max(max_ac, int(s[:, 1:].max()))
    if max(max_dc, max_ac) >= 1 << 15:
        return 0
    return 3 if max(max_dc + 7 * max_ac, 8 * max_ac) <= 5900 else 1


DYN_KINDS = [([(2, 2), (1, 1), (1, 1)], "YCbCr"), ([(1, 1), (1, 1), (1, 1)], "YCbCr"), ([(2, 1), (1, 1), (1, 1)], "YCbCr"), ([(1, 1)], "Grayscale"),
             ([(1, 2), (1, 1), (1, 1)], "YCbCr"), ([(1, 1)] * 4, "YCCK"), ([(4, 1), (1, 1), (1, 1)], "YCbCr"), ([(3, 1), (1, 1), (1, 1)], "YCbCr"),
             ([(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK"), ([(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK")]
DYN_IDS = ["420", "444", "422", "gray", "440", "ycck", "411", "311-generic", "cmyk-2211", "ycck-2212"]


@pytest.mark.parametrize("samp,ct", DYN_KINDS, ids=DYN_IDS)
def test_classes_decided_on_the_device_every_kernel(samp, ct):
    """jpgpu_batch_classify_on_device: the range statistics stay in HBM, a finalize kernel turns them into the images' classes
    in front of the pixel kernels, and ONE `_dyn` launch per kind branches per workgroup — 64 images of all three classes, every
    one equal to the oracle, and the split the device arrives at equal to the host's exact classification."""
    rng = np.random.default_rng(len(samp) * 100 + samp[0][0] * 10 + samp[0][1] + 5)
    w_, h_ = 200, 120
    cases = [_batch_case(rng, w_, h_, samp, ct, kind="tight") for _ in range(64)]
    for i in (3, 17, 63):
        cases[i] = _batch_case(rng, w_, h_, samp, ct, kind="full")
    for i in (0, 40):
        cases[i] = _batch_case(rng, w_, h_, samp, ct, kind="sane")
    descs = [J.image_desc(list(to_j(oc)), qts, w, h, ct_) for oc, qts, _c, ct_, w, h in cases]
    b = J.Batch(descs)
    try:
        for i, (oc, qts, coefs, ct_, _w, _h) in enumerate(cases):
            for c in range(len(coefs)):
                b.upload(i, c, coefs[c])
                b._check(J._native.lib().jpgpu_batch_set_range_class(b._h, i, c, 0))  # forget what upload() found out
        b.classify_on_device()
        for rep in range(2):  # (the second decode reuses statistics and tables)
            b.decode()
        b.synchronize()
        for i, (oc, qts, coefs, ct_, _w, _h) in enumerate(cases):
            assert np.array_equal(b.download(i), O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper())), i
        if b.path.startswith("fused"):
            want = [0, 0, 0]
            for oc, qts, coefs, *_ in cases:
                want[{0: 0, 1: 1, 3: 2}[_exact_image_class(qts, coefs)]] += 1
            assert b.class_counts() == tuple(want), (b.class_counts(), want)
            assert want[0] >= 3 and want[2] >= 50
        # a class set from the host afterwards takes over for that image: everything hostile -> still the oracle's pixels
        for i in range(len(cases)):
            b.set_range_hint(i, 0)
        b.decode()
        b.synchronize()
        if b.path.startswith("fused"):
            assert b.class_counts() == (64, 0, 0)
        for i in (0, 1, 17):
            oc, qts, coefs, ct_, _w, _h = cases[i]
            assert np.array_equal(b.download(i), O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper())), i
    finally:
        b.close()


@pytest.mark.parametrize("samp,ct", [DYN_KINDS[0], DYN_KINDS[1], DYN_KINDS[3], DYN_KINDS[7]], ids=["420", "444", "gray", "311-generic"])
def test_compact_upload_ranged_by_the_expansion_kernel(samp, ct):
    """jpgpu_batch_upload_compact with range_class = -1: expand_compact_kernel ranges the values while it has them in registers;
    the class drawn from its two maxima never overstates the exact one, and the pixels are the oracle's."""
    rng = np.random.default_rng(99 + len(samp))
    w_, h_ = 136, 72
    kinds = ["tight", "sparse", "full", "sane", "tight", "sparse"]
    cases = [_batch_case(rng, w_, h_, samp, ct, kind=k) for k in kinds]
    descs = [J.image_desc(list(to_j(oc)), qts, w, h, ct_) for oc, qts, _c, ct_, w, h in cases]
    b = J.Batch(descs)
    try:
        for i, (oc, qts, coefs, ct_, _w, _h) in enumerate(cases):
            for c in range(len(coefs)):
                b.upload_compact(i, c, coefs[c], classify=(i == 1))  # image 1: the host encoder's class next to the device's
        b.decode()
        b.synchronize()
        for i, (oc, qts, coefs, ct_, _w, _h) in enumerate(cases):
            assert np.array_equal(b.download(i), O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper())), i
        if b.path.startswith("fused"):
            want = [0, 0, 0]
            for i, (oc, qts, coefs, *_r) in enumerate(cases):
                cls = _exact_image_class(qts, coefs) if i == 1 else _by_product_image_class(qts, coefs)
                assert cls <= _exact_image_class(qts, coefs)
                want[{0: 0, 1: 1, 3: 2}[cls]] += 1
            assert b.class_counts() == tuple(want), (b.class_counts(), want)
        # the same components sent again with other data: the statistics start afresh (a hostile image becomes a tame one)
        oc, qts, coefs, ct_, _w, _h = cases[0]
        for c in range(len(coefs)):
            b.upload_compact(2, c, coefs[c], classify=False)
            b.set_quantization_table(2, c, qts[c])
        for c in range(len(coefs)):  # (the table change reset the class to "unknown": send once more under the new table)
            b.upload_compact(2, c, coefs[c], classify=False)
        b.decode()
        b.synchronize()
        assert np.array_equal(b.download(2), O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper()))
        if b.path.startswith("fused"):
            want = [0, 0, 0]
            for i, (oc_, qts_, coefs_, *_r) in enumerate(cases):
                cls = _exact_image_class(qts_, coefs_) if i == 1 else _by_product_image_class(*((qts, coefs) if i == 2 else (qts_, coefs_)))
                want[{0: 0, 1: 1, 3: 2}[cls]] += 1
            assert b.class_counts() == tuple(want), (b.class_counts(), want)  # image 2 left the wrap-exact class
    finally:
        b.close()


def _column_pile_case(w_, h_, samp):
    """Blocks whose first column holds ONE coefficient of |c*q| = 5000 (column sum 5000 <= 5900: the tight class), and the six
    values that pile the same column up to 35,000 without exceeding any per-coefficient maximum already seen."""
    ocomps, _ = O.make_components(w_, h_, samp)
    qts = [np.ones(64, np.uint16) for _ in ocomps]
    base, piled, idx, delta = [], [], [], []
    for c in ocomps:
        n = c.block_w * c.block_h
        a = np.zeros((n, 64), np.int16)
        a[:, 8] = 5000 * np.where(np.arange(n) % 2, -1, 1)  # row 1, column 0
        b = a.copy()
        rows = np.arange(2, 8) * 8
        b[:, rows] = a[:, 8:9]  # rows 2..7 of column 0: the same value again
        base.append(a.reshape(-1))
        piled.append(b.reshape(-1))
        ii = (np.arange(n)[:, None] * 64 + rows[None, :]).reshape(-1)
        idx.append(ii.astype(np.uint32))
        delta.append(b.reshape(-1)[ii].astype(np.int32))
    return ocomps, qts, base, piled, idx, delta


@pytest.mark.parametrize("samp,ct", [DYN_KINDS[0], DYN_KINDS[1], DYN_KINDS[3]], ids=["420", "444", "gray"])
def test_exact_column_maximum_does_not_survive_a_by_product_writer(samp, ct):
    """jpgpu_batch_classify_on_device stores an exact column maximum (RS_COL_EXACT); a later writer that ranges
    single coefficients only (here: the compact transport with an unknown class, expand_compact_kernel) can raise a column's sum
    without raising any per-coefficient maximum.  The stale column word must not keep the image in the tight class (i16 column
    outputs): pixels equal to the oracle's, and the image counted outside class 3."""
    w_, h_ = 96, 64
    ocomps, qts, base, piled, idx, delta = _column_pile_case(w_, h_, samp)
    desc = J.image_desc(list(to_j(ocomps)), qts, w_, h_, ct)
    b = J.Batch([desc, desc])
    try:
        for i in range(2):
            for c in range(len(ocomps)):
                b.upload(i, c, base[c])
                b._check(J._native.lib().jpgpu_batch_set_range_class(b._h, i, c, 0))  # forget what upload() found out
        b.classify_on_device()
        b.decode()
        b.synchronize()
        want0 = O.pixels_from_coefficients(ocomps, qts, base, w_, h_, ct.upper())
        assert np.array_equal(b.download(0), want0) and np.array_equal(b.download(1), want0)
        if b.path.startswith("fused"):
            assert b.class_counts() == (0, 0, 2)  # column sums of 5000: tight, by the scan's exact maximum
        for c in range(len(ocomps)):  # image 1 only
            b.upload_compact(1, c, piled[c], classify=False)
        b.decode()
        b.synchronize()
        assert np.array_equal(b.download(0), want0)
        assert np.array_equal(b.download(1), O.pixels_from_coefficients(ocomps, qts, piled, w_, h_, ct.upper()))
        if b.path.startswith("fused"):
            assert b.class_counts() == (0, 1, 1), b.class_counts()  # image 1: column sums of 35,000 -> sane, not tight
    finally:
        b.close()


@pytest.mark.parametrize("kind,want_cls", [("tight", 3), ("sane", 1), ("full", 0)])
def test_worker_fused_route_classifies_on_the_device(kind, want_cls):
    """The drop-in surface's fused route ranges the frame's coefficients on the device (a scan in front of the kernel, nothing
    read back) instead of running every frame wrap-exact."""
    w_, h_, samp, ct = 250, 130, [(2, 2), (1, 1), (1, 1)], "YCbCr"
    rng = np.random.default_rng(31 + want_cls)
    oc, qts, coefs, _ct, _w, _h = _batch_case(rng, w_, h_, samp, ct, kind=kind)
    comps = to_j(oc)
    want = O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct.upper())
    with J.HipWorker() as w:
        for rep in range(3):
            for i in range(3):
                w.start(J.RowData(i, comps[i], qts[i]))
                w.append_rows_contiguous(i, coefs[i], comps[i].block_height // comps[i].vertical_sampling_factor)
                w.finish_plane(i, i)
            got = w.compute_image(list(comps), None, (w_, h_), ct)
            assert w.last_path == "fused420" and np.array_equal(got, want)
            assert w.last_class == _exact_image_class(qts, coefs) == want_cls


# ---- four components with half-size ones: the strip walk W4 of csrc/fused_x4.hpp (rounds 3-4: a row kernel) ----
X4_LAYOUTS = [([(2, 2), (1, 1), (1, 1), (1, 1)], "fused420x4-2211"), ([(2, 2), (1, 1), (1, 1), (2, 2)], "fused420x4-2212")]


@pytest.mark.parametrize("size", [(65, 47), (600, 40), (16, 16), (2, 2), (13, 90), (290, 18), (577, 33), (1920, 64)], ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("ct", ["CMYK", "YCCK"])
@pytest.mark.parametrize("samp,path", X4_LAYOUTS, ids=["2211", "2212"])
@pytest.mark.parametrize("kind", ["sparse", "tight", "full"])
def test_batch_four_components_with_half_size_ones_fused(samp, path, ct, size, kind):
    """jpg-cmyk-2.jpg's layout (22 11 11 11) and YCCK with K at full size (22 11 11 22), both colour functions: the fused
    kernel (a strip walk: carry rows between MCU rows, seam rounds between segments) equals the oracle and the generic kernel pair, for every arithmetic class, several tiles / MCU rows and sizes that end inside a block / an MCU / a tile."""
    w_, h_ = size
    rng = np.random.default_rng(w_ * 13 + h_ + len(kind))
    cases = [_batch_case(rng, w_, h_, samp, ct, kind=kind) for _ in range(3)]
    outs, got_path = _run_batch(cases)
    assert got_path == path
    gen, gen_path = _run_batch(cases, flags=J._native.BATCH_FORCE_GENERIC)
    assert gen_path == "generic"
    for (oc, qts, coefs, ct_, _w, _h), got, g2 in zip(cases, outs, gen):
        want = O.pixels_from_coefficients(oc, qts, coefs, w_, h_, ct_.upper())
        assert np.array_equal(got, want)
        assert np.array_equal(g2, want)


@pytest.mark.parametrize("samp,mode,ct,path", [([(2, 2), (1, 1), (1, 1), (1, 1)], "cmyk", "CMYK", "fused420x4-2211"),
                                               ([(2, 2), (1, 1), (1, 1), (2, 2)], "ycck", "YCCK", "fused420x4-2212")], ids=["cmyk-2211", "ycck-2212"])
def test_batch_1080p_four_components_full_size(samp, mode, ct, path):
    w_, h_ = 1920, 1080
    ocomps, _ = O.make_components(w_, h_, samp)
    lum, chr_ = synth.quality_tables(85)
    qts = [lum] * 4 if mode == "cmyk" else [lum, chr_, chr_, lum]
    base = synth.coefficients_from_rgb(synth.synthetic_rgb(w_, h_), to_j(ocomps), mode, qts)
    outs, got_path = _run_batch([(ocomps, qts, base, ct, w_, h_)] * 4)
    assert got_path == path
    want = O.pixels_from_coefficients(ocomps, qts, base, w_, h_, ct.upper())
    for got in outs:
        assert np.array_equal(got, want)


# ---- reduced-size decodes in one launch (csrc/fused_scaled.hpp) -------------------------------------------
def _scaled_case(rng, w_, h_, samp, ct, scale, kind="sparse"):
    ocomps, _ = O.make_components(w_, h_, samp, dct_scale=scale)
    if kind == "sparse":
        qts = [rng.integers(1, 64, 64).astype(np.uint16) for _ in ocomps]
        coefs = [synth.sparse_coefficients(rng, c.block_w * c.block_h, amp=64, dc_amp=500) for c in ocomps]
    else:
        qts = [rng.integers(1, 65536, 64).astype(np.uint16) for _ in ocomps]
        coefs = [rng.integers(-32768, 32768, c.block_w * c.block_h * 64).astype(np.int16) for c in ocomps]
    ow, oh = J.scaled_output_size(w_, h_, scale)
    return ocomps, qts, coefs, ct, ow, oh


SCALED_KINDS = {
    "420": ([(2, 2), (1, 1), (1, 1)], "YCbCr", "fused420-s%d"), "444": ([(1, 1), (1, 1), (1, 1)], "YCbCr", "fused444-s%d"),
    "444rgb": ([(1, 1), (1, 1), (1, 1)], "RGB", "fused444-s%d"), "422": ([(2, 1), (1, 1), (1, 1)], "YCbCr", "fused422-s%d"),
    "440": ([(1, 2), (1, 1), (1, 1)], "YCbCr", "fusedscaled-s%d"), "411": ([(4, 1), (1, 1), (1, 1)], "YCbCr", "fusedscaled-s%d"),
    "311": ([(3, 1), (1, 1), (1, 1)], "YCbCr", "fusedscaled-s%d"), "gray": ([(1, 1)], "Grayscale", "fusedgray-s%d"),
    "cmyk": ([(1, 1)] * 4, "CMYK", "fusedscaled-s%d"), "cmyk2211": ([(2, 2), (1, 1), (1, 1), (1, 1)], "CMYK", "fusedscaled-s%d"),
    "ycck2212": ([(2, 2), (1, 1), (1, 1), (2, 2)], "YCCK", "fusedscaled-s%d"), "none": ([(1, 1)] * 3, "None", "fused444-s%d"),
}
SCALED_SIZES = [(64, 48), (33, 17), (1, 1), (3, 5), (250, 130), (1930, 40), (9, 300), (640, 480), (1025, 24)]


@pytest.mark.parametrize("key", sorted(SCALED_KINDS))
@pytest.mark.parametrize("scale", [4, 2, 1])
@pytest.mark.parametrize("kind", ["sparse", "full"])
def test_batch_reduced_size_decodes_take_the_fused_scaled_kernel(key, scale, kind):
    """Images of one layout but different sizes at one reduced scale: ONE launch (coefficients -> pixels, planes in LDS), bit-exact
    against the oracle's Decoder::scale decode and against the generic pair of kernels (JPGPU_BATCH_FORCE_GENERIC)."""
    samp, ct, path = SCALED_KINDS[key]
    rng = np.random.default_rng(len(key) * 313 + scale * 7 + len(kind))
    cases = [_scaled_case(rng, w_, h_, samp, ct, scale, kind) for (w_, h_) in SCALED_SIZES]
    outs, got_path = _run_batch(cases)
    assert got_path == path % scale
    gen, gen_path = _run_batch(cases, flags=J._native.BATCH_FORCE_GENERIC)
    assert gen_path == "generic"
    for i, ((oc, qts, coefs, ct_, ow, oh), got, g2) in enumerate(zip(cases, outs, gen)):
        want = O.pixels_from_coefficients(oc, qts, coefs, ow, oh, ct_.upper())
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (i, ow, oh, bad.size, bad[:16].tolist(), got[bad[:16]].tolist(), want[bad[:16]].tolist())
        assert np.array_equal(g2, want), (ow, oh)


def test_batch_reduced_and_full_size_images_of_many_kinds_in_one_batch():
    """Scales 8 / 4 / 2 / 1, several layouts, one batch: fused launch groups for the full-size images, ONE fused-scaled launch per
    reduced scale present, nothing left for the generic pair — path "mixed", every image the oracle's."""
    rng = np.random.default_rng(2024)
    cases = []
    for i, (w_, h_) in enumerate([(250, 130), (64, 48), (1000, 70), (33, 17), (500, 333), (17, 1080)]):
        for scale in (8, 4, 2, 1):
            samp, ct, _p = list(SCALED_KINDS.values())[(i + scale) % len(SCALED_KINDS)]
            cases.append(_scaled_case(rng, w_, h_, samp, ct, scale))
    outs, path = _run_batch(cases)
    assert path == "mixed"
    for (oc, qts, coefs, ct_, ow, oh), got in zip(cases, outs):
        assert np.array_equal(got, O.pixels_from_coefficients(oc, qts, coefs, ow, oh, ct_.upper())), (ow, oh, ct_)


@pytest.mark.parametrize("scale", [4, 2, 1])
def test_batch_1080p_420_reduced_size_full_batch(scale):
    """The bench's reduced-size workloads at their own size: 1920x1080 4:2:0 at 4/8, 2/8, 1/8 (bench.py 1080p-420-scaleN)."""
    w_, h_ = 1920, 1080
    samp = [(2, 2), (1, 1), (1, 1)]
    ocomps, _ = O.make_components(w_, h_, samp, dct_scale=scale)
    lum, chr_ = synth.quality_tables(85)
    qts = [lum, chr_, chr_]
    comps_full, _ = O.make_components(w_, h_, samp)
    coefs = synth.coefficients_from_rgb(synth.synthetic_rgb(w_, h_), to_j(comps_full), "ycbcr", qts)
    ow, oh = J.scaled_output_size(w_, h_, scale)
    want = O.pixels_from_coefficients(ocomps, qts, coefs, ow, oh, "YCBCR")
    desc = J.image_desc(list(to_j(ocomps)), qts, ow, oh, "YCbCr")
    b = J.Batch([desc] * 6)
    try:
        for i in range(6):
            for c in range(3):
                b.upload(i, c, coefs[c])
        b.decode()
        b.synchronize()
        assert b.path == "fused420-s%d" % scale
        for i in range(6):
            assert np.array_equal(b.download(i), want), i
    finally:
        b.close()


def _hip():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return hip


@pytest.mark.parametrize("scale", [8, 4, 2, 1])
def test_every_output_byte_is_written_whatever_the_arena_held(scale):
    """Kernels that write a pixel arena in pieces (tiles, strips, rings) must cover it exactly: the batch decodes into a caller's
    arena pre-filled with one pattern, then another — bytes a kernel forgets keep the pattern and show up against the oracle, bytes
    between the images (and behind the last one) must keep it.  Mixed layouts and sizes in one batch, every reduced scale."""
    import ctypes as C
    hip = _hip()
    rng = np.random.default_rng(900 + scale)
    layouts = [SCALED_KINDS[k] for k in ("420", "444", "422", "440", "gray", "cmyk2211", "411")]
    sizes = [(250, 130), (1930, 40), (33, 17), (640, 480), (9, 300), (1025, 24), (64, 48)]
    cases = [_scaled_case(rng, w_, h_, samp, ct, scale) for (samp, ct, _p), (w_, h_) in zip(layouts, sizes)]
    descs = [J.image_desc(list(to_j(oc)), qts, ow, oh, ct_) for oc, qts, _c, ct_, ow, oh in cases]
    b = J.Batch(descs, flags=J._native.BATCH_EXTERNAL_BUFFERS)
    coef, out = C.c_void_p(), C.c_void_p()
    nco, nout = b.coef_arena_bytes(), b.out_arena_bytes()
    assert hip.hipMalloc(C.byref(coef), nco) == 0 and hip.hipMalloc(C.byref(out), nout + 4096) == 0
    try:
        b.bind(coef.value, out.value)
        for i, (oc, qts, coefs, ct_, ow, oh) in enumerate(cases):
            for c in range(len(oc)):
                b.upload(i, c, coefs[c])
        for pattern in (0xA5, 0x3C):
            assert hip.hipMemset(out, pattern, nout + 4096) == 0
            b.decode()
            b.synchronize()
            host = np.empty(nout + 4096, np.uint8)
            assert hip.hipMemcpy(host.ctypes.data, out, nout + 4096, 2) == 0
            covered = np.zeros(nout + 4096, bool)
            for i, (oc, qts, coefs, ct_, ow, oh) in enumerate(cases):
                want = O.pixels_from_coefficients(oc, qts, coefs, ow, oh, ct_.upper())
                off = b.out_offset(i)
                assert b.out_bytes(i) == want.size
                got = host[off: off + want.size]
                bad = np.nonzero(got != want)[0]
                assert bad.size == 0, (scale, hex(pattern), i, ow, oh, bad.size, bad[:16].tolist(), got[bad[:16]].tolist(), want[bad[:16]].tolist())
                covered[off: off + want.size] = True
            assert (host[~covered] == pattern).all(), "a kernel wrote outside the images"
    finally:
        b.close()
        hip.hipFree(coef)
        hip.hipFree(out)
