"""End to end on the MI355X: jpeg_decoder_amd.Decoder (C++ front-end + HIP kernels through the C ABI)
on the reference's own test and bench images, byte-exact against the oracle / golden hashes."""
import hashlib
import os

import numpy as np
import pytest

import oracle as O
import refimages as R

pytestmark = pytest.mark.gpu
J = None


@pytest.fixture(scope="module", autouse=True)
def _load():
    global J
    import jpeg_decoder_amd as pkg
    J = pkg
    assert J.device_count() >= 1


@pytest.mark.parametrize("rel", R.reftest_files())
def test_reftest_decode(rel):
    """tests/reftest/mod.rs: Decoder::new(file).decode() vs the sibling PNG (<= 3) — and exact vs oracle."""
    path = os.path.join(R.REFTEST, rel)
    data = open(path, "rb").read()
    d = J.Decoder(data)
    got = d.decode()
    od = O.decode(data)
    assert np.array_equal(got, od.pixels)
    i = d.info()
    assert (i.width, i.height, i.pixel_format) == (od.width, od.height, od.pixel_format)
    assert R.max_diff_vs_png(got, od.ncomp, os.path.splitext(path)[0] + ".png") <= 3


@pytest.mark.parametrize("key", sorted(R.golden_hashes()))
def test_golden_sha256(key):
    rel, _, scale = key.partition("@")
    d = J.Decoder(open(os.path.join(R.GOLDEN, rel), "rb").read())
    if scale:
        w, h = (int(v) for v in scale.split("x"))
        assert d.scale(w, h) == (w, h)
    assert hashlib.sha256(d.decode().tobytes()).hexdigest() == R.golden_hashes()[key]


def test_read_info_then_decode_same_as_decode():
    # tests/lib.rs:34-50
    data = open(os.path.join(R.REFTEST, "mozilla", "jpg-progressive.jpg"), "rb").read()
    a = J.Decoder(data)
    ref = a.decode()
    b = J.Decoder(data)
    b.read_info()
    info = b.info()
    got = b.decode()
    assert info == b.info() == a.info()
    assert np.array_equal(got, ref)


def test_color_transform_override():
    data = open(os.path.join(R.GOLDEN, "benches", "tower.jpg"), "rb").read()
    for ct in ("RGB", "YCbCr", "None"):
        d = J.Decoder(data)
        d.set_color_transform(ct)
        assert np.array_equal(d.decode(), O.decode(data, color_transform=ct.upper()).pixels), ct
    d = J.Decoder(data)
    d.set_color_transform("CMYK")
    with pytest.raises(J.FormatError):
        d.decode()


def test_decode_batch_of_files():
    names = ["benches/tower.jpg", "reftest/mjpeg.jpg", "reftest/rgb.jpg", "benches/tower_grayscale.jpg",
             "reftest/mozilla/jpg-cmyk-2.jpg", "benches/tower_progressive.jpg", "reftest/non-interleaved-mcu.jpg"]
    files = [open(os.path.join(R.GOLDEN, n), "rb").read() for n in names]
    out = J.decode_batch(files)
    for n, f, (info, px) in zip(names, files, out):
        assert np.array_equal(px, O.decode(f).pixels), n
    same = J.decode_batch([files[0]] * 6)  # same geometry -> fused kernels
    for info, px in same:
        assert hashlib.sha256(px.tobytes()).hexdigest() == R.golden_hashes()["benches/tower.jpg"]


def test_hostile_files_never_crash_the_gpu_path():
    """tests/crashtest/mod.rs on the full decode(): errors allowed, crashes not; files that decode
    must still match the oracle (wrap-around IDCT semantics on hostile coefficients)."""
    import glob
    files = sorted(glob.glob(os.path.join(R.GOLDEN, "crashtest", "*.jpg")) +
                   glob.glob(os.path.join(R.GOLDEN, "crashtest", "imagetestsuite", "*.jpg")))
    n_ok = 0
    for f in files:
        data = open(f, "rb").read()
        try:
            want = O.decode(data).pixels
        except O.OracleError as e:
            with pytest.raises(J.Error) as pe:
                J.Decoder(data).decode()
            assert pe.value.kind == e.kind, f
            continue
        assert np.array_equal(J.Decoder(data).decode(), want), f
        n_ok += 1
    assert n_ok >= 5


def test_large_sequential_images_take_the_device_entropy_route_with_the_same_results(monkeypatch):
    """Decoder.decode() of sequential images of 1.5 MP and more goes through a one-image device-entropy pipeline call;
    results, metadata, repeated calls, decode_coefficients() afterwards, options (scale / colour transform: ordinary route)
    and damaged streams behave as on the ordinary route."""
    pytest.importorskip("PIL")
    import io
    from PIL import Image
    import synth
    big = open(os.path.join(R.GOLDEN, "benches", "large_image.jpg"), "rb").read()  # 2268x1512: above the threshold
    want = O.decode(big).pixels
    d = J.Decoder(big)
    a = d.decode()
    b = d.decode()  # (the first call kept no copy: decoded again)
    assert np.array_equal(a, want) and np.array_equal(b, want)
    i = d.info()
    assert (i.width, i.height) == (2268, 1512)
    desc, coefs = d.decode_coefficients()  # needs a working front-end again
    assert desc.ncomp == 3 and all(c.size for c in coefs)
    monkeypatch.setenv("JPGPU_DECODER_NO_DEVICE_ENTROPY", "1")
    assert np.array_equal(J.Decoder(big).decode(), want)
    monkeypatch.delenv("JPGPU_DECODER_NO_DEVICE_ENTROPY")
    d2 = J.Decoder(big)
    d2.scale(600, 400)  # options switch to the ordinary route
    assert np.array_equal(d2.decode(), O.decode(big, scale_to=(600, 400)).pixels)
    for (w, h, sub, kw) in [(1600, 1200, "4:2:0", {}), (2000, 1000, "4:4:4", {"restart_marker_rows": 4}), (1400, 1100, "4:2:2", {"progressive": True})]:
        buf = io.BytesIO()
        Image.fromarray(synth.synthetic_rgb(w, h, seed=w)).save(buf, format="JPEG", quality=85, subsampling=sub, **kw)
        data = buf.getvalue()
        assert np.array_equal(J.Decoder(data).decode(), O.decode(data).pixels), (w, h, sub)
        cut = data[: len(data) * 2 // 3]  # truncated: same outcome as the oracle's (pixels or the same kind of error)
        try:
            wantc = O.decode(cut).pixels
        except O.OracleError as e:
            with pytest.raises(J.Error) as ei:
                J.Decoder(cut).decode()
            assert ei.value.kind == e.kind
        else:
            assert np.array_equal(J.Decoder(cut).decode(), wantc)


def test_trim_caches_between_decodes():
    """jpgpu_trim_caches releases the idle workers / pipelines / host buffers the decoders borrow from; decoding afterwards
    simply builds new ones."""
    from jpeg_decoder_amd import _native as N
    data = open(os.path.join(R.GOLDEN, "benches", "tower.jpg"), "rb").read()
    want = O.decode(data).pixels
    assert np.array_equal(J.Decoder(data).decode(), want)
    N.lib().jpgpu_trim_caches()
    N.lib().jpgpu_trim_caches()
    assert np.array_equal(J.Decoder(data).decode(), want)
    prog = open(os.path.join(R.GOLDEN, "benches", "tower_progressive.jpg"), "rb").read()
    assert np.array_equal(J.Decoder(prog).decode(), O.decode(prog).pixels)
    N.lib().jpgpu_trim_caches()
    assert np.array_equal(J.Decoder(prog).decode(), O.decode(prog).pixels)


@pytest.mark.parametrize("name", R.anchor_files())
def test_anchor_440_411_bit_exact_and_within_tolerance_of_libjpeg_turbo(name):
    """The 4:4:0 / 4:1:1 anchor files (tests/golden/anchor: decoded by libjpeg-turbo): GPU == oracle byte for byte, through the
    Decoder (Worker route: fused440 / fusedgen kernels) and through a batch of all of them at once, and within 3 of the
    external decode."""
    path = os.path.join(R.ANCHOR, name)
    data = open(path, "rb").read()
    od = O.decode(data)
    got = J.Decoder(data).decode()
    assert np.array_equal(got, od.pixels)
    assert R.max_diff_vs_png(got, od.ncomp, os.path.splitext(path)[0] + ".png") <= 3


def test_anchor_files_through_the_pipeline():
    files = [open(os.path.join(R.ANCHOR, n), "rb").read() for n in R.anchor_files()]
    assert len(files) >= 6
    p = J.Pipeline()
    try:
        for dev in (True, False):
            out = p.decode(files, download=True, device_entropy=dev)
            for data, got in zip(files, out):
                assert not isinstance(got, Exception), got
                assert np.array_equal(got, O.decode(data).pixels)
    finally:
        p.close()
