"""The 4:2:0 walk that reads the chunk decoder's entry lists itself (csrc/fused_entries.hpp, JPGPU_PIPE_ENTRY_PIXELS): through
jpgpu_pipeline_decode every image must come out as the oracle decodes it — strips with and without halo MCUs, seams between
segments, runs that begin in the middle of a chunk's list, blocks that straddle chunks, damaged streams, coefficients outside the
class the walk runs on trust."""
import io
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
    assert J.device_count() >= 1


def _jpeg(w, h, quality=85, seed=1, kind="photo", subsampling=2):
    from PIL import Image
    if kind == "flat":
        rgb = np.full((h, w, 3), 90, np.uint8)
        rgb[h // 3:, w // 2:] = (200, 40, 120)
    elif kind == "noise":
        rgb = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
    else:
        rgb = synth.synthetic_rgb(w, h, seed=seed)
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="JPEG", quality=quality, subsampling=subsampling)
    return buf.getvalue()


def _decode_and_check(p, files, expect_entry, expect_rejected=0):
    out = p.decode(files)
    t = p.timings()
    assert t["images_entry_pixels"] == expect_entry, t
    assert t["images_device_rejected"] == expect_rejected, t  # (pixels equal to the oracle's are not enough: a host re-decode makes them too)
    for i, (f, got) in enumerate(zip(files, out)):
        try:
            want = O.decode(f).pixels
        except O.OracleError as e:
            assert isinstance(got, J.Error) and got.kind == e.kind, (i, got, e)
            continue
        assert not isinstance(got, Exception), (i, got)
        assert np.array_equal(got, want), (i, got.shape, int(np.abs(got.astype(int) - want.astype(int)).max()))
    return t


SIZES = [(16, 16), (17, 33), (48, 48), (100, 75), (333, 200), (640, 480), (700, 260), (1400, 96), (1920, 1080)]


@pytest.mark.parametrize("knobs", [{}, {"JPGPU_S420_TX": "3", "JPGPU_S420_SEG": "2"}, {"JPGPU_S420_TX": "1", "JPGPU_S420_SEG": "1"},
                                   {"JPGPU_S420_SEG": "3"}, {"JPGPU_SYNC_BLOCKS": "6", "JPGPU_SYNC_MIN_SHIFT": "7", "JPGPU_S420_TX": "5"}])
def test_entry_lists_feed_the_walk_sizes_and_knobs(monkeypatch, knobs):
    """Every size twice in a call of its own geometry (uniform launch groups) and all of them in one call (mixed), at strip widths
    and segment lengths that put halos, seams and runs starting inside a list on small images; tiny chunks make blocks straddle them."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    p = J.Pipeline(threads=4)
    allf = []
    for i, (w, h) in enumerate(SIZES):
        if knobs and w * h > 700 * 520:
            continue
        files = [_jpeg(w, h, seed=10 + i), _jpeg(w, h, quality=97, seed=20 + i), _jpeg(w, h, quality=30, seed=30 + i, kind="flat")]
        if w * h <= 640 * 480:
            files.append(_jpeg(w, h, quality=92, seed=40 + i, kind="noise"))
        _decode_and_check(p, files, len(files))
        allf += files[:2]
    _decode_and_check(p, allf, len(allf))
    p.close()


def test_entry_lists_off_by_knob_and_other_layouts_keep_the_expansion(monkeypatch):
    p = J.Pipeline(threads=4)
    f420 = [_jpeg(320, 240, seed=s) for s in range(6)]
    f444 = [_jpeg(320, 240, seed=s, subsampling=0) for s in range(3)]
    f422 = [_jpeg(320, 240, seed=s, subsampling=1) for s in range(3)]
    _decode_and_check(p, f420 + f444 + f422, len(f420))  # (one call, three launch groups: only the 4:2:0 one reads lists)
    monkeypatch.setenv("JPGPU_PIPE_ENTRY_PIXELS", "0")
    _decode_and_check(p, f420 + f444, 0)
    monkeypatch.delenv("JPGPU_PIPE_ENTRY_PIXELS")
    _decode_and_check(p, f420, len(f420))
    p.close()


@pytest.mark.parametrize("knobs", [{}, {"JPGPU_S420_TX": "3", "JPGPU_S420_SEG": "2"}, {"JPGPU_SYNC_BLOCKS": "6", "JPGPU_SYNC_MIN_SHIFT": "7", "JPGPU_S420_TX": "5"}])
def test_entry_lists_of_streams_with_restart_markers(monkeypatch, knobs):
    """Restart segments have chunk slots of their own and block numbers that run on across them; a segment's first chunk continues
    nothing, and what its last chunk makes of the bits behind the segment's last block is nobody's.  Intervals of whole MCU rows, of a
    few MCUs (runs that span segments) and of one MCU."""
    from PIL import Image
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    files = []
    sizes = [(320, 240, {"restart_marker_rows": 1}), (320, 240, {"restart_marker_rows": 3}), (320, 240, {"restart_marker_blocks": 7}),
             (700, 260, {"restart_marker_blocks": 5}), (100, 75, {"restart_marker_blocks": 1}), (640, 480, {"restart_marker_rows": 2})]
    if not knobs:
        sizes += [(1920, 1080, {"restart_marker_rows": 1}), (1920, 1080, {"restart_marker_blocks": 50})]
    for i, (w, h, kw) in enumerate(sizes):
        for q in (85, 40):
            buf = io.BytesIO()
            Image.fromarray(synth.synthetic_rgb(w, h, seed=60 + i)).save(buf, format="JPEG", quality=q, subsampling=2, **kw)
            files.append(buf.getvalue())
    p = J.Pipeline(threads=4)
    _decode_and_check(p, files, len(files))
    _decode_and_check(p, [files[0]] * 5 + [_jpeg(320, 240, seed=1)] * 3, 8)
    p.close()


def test_entry_lists_damaged_streams_and_the_reference_fixtures():
    """Truncated and corrupted scans among good ones: whoever the device decoder refuses is decoded by the host, its neighbours keep the
    pixels the walk made; the crate's own 4:2:0 fixtures."""
    rng = np.random.default_rng(7)
    good = [_jpeg(400, 300, seed=s) for s in range(8)]
    files = []
    for i, g in enumerate(good):
        files.append(g)
        b = bytearray(g)
        if i % 2:
            del b[len(b) * (2 + i % 3) // 5:]
        else:
            for _ in range(3):
                b[int(rng.integers(len(b) // 2, len(b) - 2))] ^= 1 << int(rng.integers(0, 8))
        files.append(bytes(b))
    p = J.Pipeline(threads=4)
    out = p.decode(files)
    t = p.timings()
    assert t["images_entry_pixels"] >= len(good)
    for i, (f, got) in enumerate(zip(files, out)):
        try:
            want = O.decode(f).pixels
        except O.OracleError as e:
            assert isinstance(got, J.Error) and got.kind == e.kind, (i, got, e)
            continue
        assert not isinstance(got, Exception) and np.array_equal(got, want), i
    import glob
    names = sorted(glob.glob(os.path.join(R.GOLDEN, "reftest", "**", "*.jp*g"), recursive=True))
    fixtures = [open(n, "rb").read() for n in names]
    out = p.decode(fixtures)
    assert p.timings()["images_entry_pixels"] >= 3
    for n, f, got in zip(names, fixtures, out):
        try:
            want = O.decode(f).pixels
        except O.OracleError as e:
            assert isinstance(got, J.Error) and got.kind == e.kind, n
            continue
        assert not isinstance(got, Exception) and np.array_equal(got, want), n
    p.close()


def _with_quantization_values(data, value):
    """the file with every 8-bit quantization table entry set to `value` (the coefficients stay: their products leave the sane class)"""
    b = bytearray(data)
    i = 2
    while i + 4 <= len(b) and b[i] == 0xFF:
        m, ln = b[i + 1], (b[i + 2] << 8) | b[i + 3]
        if m == 0xDB:
            j = i + 4
            while j < i + 2 + ln:
                assert b[j] >> 4 == 0
                b[j + 1:j + 65] = bytes([value]) * 64
                j += 65
        if m == 0xDA:
            break
        i += 2 + ln
    return bytes(b)


def test_entry_lists_coefficients_outside_the_sane_class_go_back_to_the_host():
    """The walk runs the 'sane' arithmetic on trust and checks every coefficient it scatters: a file whose products reach 2^15 is flagged
    (status bit 9), decoded by the host with the wrap-exact kernels — and its neighbours of the same sub-batch keep their pixels."""
    base = [_jpeg(256, 192, quality=100, seed=s) for s in range(3)]
    hostile = [_with_quantization_values(b, 255) for b in base]
    tame = [_jpeg(256, 192, seed=50 + s) for s in range(5)]
    files = [tame[0], hostile[0], tame[1], tame[2], hostile[1], tame[3], hostile[2], tame[4]]
    p = J.Pipeline(threads=4)
    for _ in range(2):  # (the second call reuses the batch: the flags of the first must not stick)
        out = p.decode(files)
        t = p.timings()
        assert t["images_entry_pixels"] == len(files) and t["images_device_rejected"] == len(hostile), t
        for i, (f, got) in enumerate(zip(files, out)):
            assert np.array_equal(got, O.decode(f).pixels), i
    _decode_and_check(p, tame, len(tame))
    p.close()


def test_entry_lists_many_files_every_image_checked():
    """1,024 files of four geometries in one call (several sub-batches per stream: the lists live in a scratch block shared by the
    sub-batches of a stream — the walk of one must have read them before the next one's passes overwrite them)."""
    kinds = [(640, 360), (512, 512), (800, 200), (96, 400)]
    distinct = [[_jpeg(w, h, seed=100 + 7 * k + s, quality=70 + 5 * s) for s in range(4)] for k, (w, h) in enumerate(kinds)]
    want = [[O.decode(f).pixels for f in fs] for fs in distinct]
    files, idx = [], []
    for i in range(1024):
        k, s = i % 4, (i // 4) % 4
        files.append(distinct[k][s])
        idx.append((k, s))
    p = J.Pipeline(threads=8)
    for _ in range(2):
        out = p.decode(files)
        assert p.timings()["images_entry_pixels"] == len(files) and p.timings()["images_device_rejected"] == 0
        for (k, s), got in zip(idx, out):
            assert np.array_equal(got, want[k][s]), (k, s)
    p.close()


def test_entry_lists_of_2160p_frames_six_strips():
    """3840x2160 (BASELINE configs[2]'s frame): six strips of 40 MCUs, 135 MCU rows, lists of a few thousand chunks."""
    files = [_jpeg(3840, 2160, quality=q, seed=70 + q) for q in (85, 60)]
    p = J.Pipeline(threads=4)
    _decode_and_check(p, files * 2, 4)
    p.close()
