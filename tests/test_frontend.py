"""Host front-end (C++ marker parser + Huffman / progressive decoder behind include/jpgpu_decoder.h)
against the oracle, without a GPU: what it hands to the Worker boundary must be identical, and it
must fail exactly where the reference fails (tests/crashtest/mod.rs: never crash; errors allowed).
Also the API / metadata tests of the reference's tests/lib.rs:34-170."""
import glob
import os

import numpy as np
import pytest

import oracle as O
import refimages as R

import jpeg_decoder_amd as J

ALL_GOOD = sorted(glob.glob(os.path.join(R.REFTEST, "*.jpg")) + glob.glob(os.path.join(R.REFTEST, "mozilla", "*.jpg")) +
                  glob.glob(os.path.join(R.GOLDEN, "benches", "*.jpg")) + glob.glob(os.path.join(R.ANCHOR, "*.jpg")))
HOSTILE = sorted(glob.glob(os.path.join(R.GOLDEN, "crashtest", "*.jpg")) +
                 glob.glob(os.path.join(R.GOLDEN, "crashtest", "imagetestsuite", "*.jpg")))


def _host_decode(data, scale_to=None, ct=None):
    d = J.Decoder(data, device=-1)  # host-only object: no GPU needed
    if scale_to:
        d.scale(*scale_to)
    if ct is not None:
        d.set_color_transform(ct)
    return d, d.decode_coefficients()


def _same_as_oracle(data, **kw):
    od = O.decode(data, keep_intermediates=True, scale_to=kw.get("scale_to"))
    d, (desc, coefs) = _host_decode(data, **kw)
    assert (desc.ncomp, desc.out_w, desc.out_h, desc.color_transform) == (od.ncomp, od.width, od.height, od.color_transform)
    for c in range(od.ncomp):
        oc, pc = od.components[c], desc.components[c]
        assert (oc.h, oc.v, oc.dct_scale, oc.size_w, oc.size_h, oc.block_w, oc.block_h) == (
            pc.horizontal_sampling_factor, pc.vertical_sampling_factor, pc.dct_scale, pc.size_width, pc.size_height,
            pc.block_width, pc.block_height)
        assert list(desc.quantization_tables[c]) == list(od.qtables[c])
        assert np.array_equal(coefs[c], od.coefs[c]), c
    i = d.info()
    assert (i.width, i.height, i.pixel_format, i.coding_process) == (od.width, od.height, od.pixel_format, od.coding_process)


@pytest.mark.parametrize("path", ALL_GOOD, ids=lambda p: os.path.relpath(p, R.GOLDEN))
def test_boundary_feed_matches_oracle(path):
    data = open(path, "rb").read()
    try:
        O.decode(data)
    except O.OracleError as e:  # jpg-size-6x6.jpg is a PNG
        with pytest.raises(J.Error) as pe:
            _host_decode(data)
        assert pe.value.kind == e.kind
        return
    _same_as_oracle(data)


@pytest.mark.parametrize("req", [(250, 167), (125, 84), (63, 42), (500, 333)])
def test_scaled_feed_matches_oracle(req):
    _same_as_oracle(open(os.path.join(R.REFTEST, "rgb.jpg"), "rb").read(), scale_to=req)


@pytest.mark.parametrize("path", HOSTILE, ids=lambda p: os.path.basename(p)[:24])
def test_crash_corpus_same_outcome_as_oracle(path):
    data = open(path, "rb").read()
    try:
        od = O.decode(data, keep_intermediates=True)
        okind = "Ok"
    except O.OracleError as e:
        od, okind = None, e.kind
    try:
        _, (desc, coefs) = _host_decode(data)
        pkind = "Ok"
    except J.Error as e:
        pkind = e.kind
    assert pkind == okind
    if od is not None:
        for c in range(od.ncomp):
            assert np.array_equal(coefs[c], od.coefs[c])


def test_read_info_is_idempotent_and_precedes_decode():
    # tests/lib.rs:34-50
    data = open(os.path.join(R.REFTEST, "mozilla", "jpg-progressive.jpg"), "rb").read()
    d = J.Decoder(data, device=-1)
    assert d.info() is None
    d.read_info()
    i1 = d.info()
    d.read_info()
    assert d.info() == i1 == J.ImageInfo(32, 32, "RGB24", "DctProgressive")
    desc, _ = d.decode_coefficients()
    assert (desc.out_w, desc.out_h) == (32, 32)


def test_scale_chooses_reference_sizes():
    data = open(os.path.join(R.REFTEST, "rgb.jpg"), "rb").read()
    for req, want in (((500, 333), (500, 333)), ((250, 167), (250, 167)), ((125, 84), (125, 84)), ((63, 42), (63, 42)),
                      ((64, 43), (125, 84)), ((1, 1), (63, 42))):
        assert J.Decoder(data, device=-1).scale(*req) == want


def test_icc_profiles():
    # tests/lib.rs:52-137
    def profile(path):
        d = J.Decoder(open(path, "rb").read(), device=-1)
        d.decode_coefficients()
        return d.icc_profile()

    p = profile(os.path.join(R.REFTEST, "mozilla", "jpg-srgb-icc.jpg"))
    assert p[36:40] == b"acsp"
    p = profile(os.path.join(R.GOLDEN, "icc", "icc_chunk_order.jpeg"))
    assert len(p) == 254 and list(p) == list(range(1, 255))
    for name in ("icc_chunk_seq_no_0", "icc_chunk_double_seq_no", "icc_chunk_count_mismatch", "icc_missing_chunk"):
        assert profile(os.path.join(R.GOLDEN, "icc", name + ".jpeg")) is None


def test_exif_and_xmp():
    # tests/lib.rs:139-170
    d = J.Decoder(open(os.path.join(R.REFTEST, "ycck.jpg"), "rb").read(), device=-1)
    d.decode_coefficients()
    assert d.exif_data()[:8] == b"\x49\x49\x2A\x00\x08\x00\x00\x00"
    assert d.xmp_data()[:9] == b"<?xpacket"


def test_decode_without_a_device_fails_loudly():
    d = J.Decoder(open(os.path.join(R.REFTEST, "rgb.jpg"), "rb").read(), device=-1)
    with pytest.raises(J.Error):
        d.decode()  # the pixel pipeline has no CPU fallback


def test_max_decoding_buffer_size():
    d = J.Decoder(open(os.path.join(R.REFTEST, "rgb.jpg"), "rb").read(), device=-1)
    d.set_max_decoding_buffer_size(1000)
    with pytest.raises(J.FormatError, match="exceeds maximum"):
        d.decode_coefficients()


FUZZ_SEEDS = ["reftest/mozilla/jpg-size-33x33.jpg", "reftest/mozilla/jpg-progressive.jpg", "reftest/restarts.jpg",
              "reftest/non-interleaved-mcu.jpg", "reftest/mozilla/jpg-gray.jpg", "reftest/16bit-qtables.jpg", "benches/tower.jpg",
              "benches/tower_progressive.jpg", "reftest/progressive3.jpg"]  # (refinement scans through the non-zero bitmaps)


@pytest.mark.parametrize("rel", FUZZ_SEEDS)
def test_mutated_streams_same_outcome_as_oracle(rel):
    """Seeded mutation fuzzing: corrupt bytes of valid files (entropy data, Huffman tables, headers) and require the
    same outcome as the oracle — same error kind, or the very same coefficients.  The front-end's wide lookup tables
    and 8-byte refill are caches of the reference's bit-serial procedure; damaged streams are where a cache that is
    not exact (e.g. on malformed Huffman tables, or past the end of a band) would show."""
    base = bytearray(open(os.path.join(R.GOLDEN, rel), "rb").read())
    rng = np.random.default_rng(len(base))
    n_ok = n_err = 0
    for trial in range(120):
        data = bytearray(base)
        lo = 2 if trial % 3 else max(2, len(data) // 3)   # two thirds of the trials may also hit the headers / tables
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(lo, len(data)))
            mode = int(rng.integers(0, 4))
            if mode == 0:
                data[pos] ^= 1 << int(rng.integers(0, 8))
            elif mode == 1:
                data[pos] = int(rng.integers(0, 256))
            elif mode == 2:
                data[pos] = 0xFF
            else:
                del data[pos: pos + int(rng.integers(1, 5))]
        data = bytes(data)
        try:
            od = O.decode(data, keep_intermediates=True)
            okind = "Ok"
        except O.OracleError as e:
            od, okind = None, e.kind
        try:
            _, (desc, coefs) = _host_decode(data)
            pkind = "Ok"
        except J.Error as e:
            pkind = e.kind
        assert pkind == okind, (rel, trial)
        if od is not None:
            n_ok += 1
            for c in range(od.ncomp):
                assert np.array_equal(coefs[c], od.coefs[c]), (rel, trial, c)
        else:
            n_err += 1
    assert n_ok >= 10 and n_err >= 5, (n_ok, n_err)
