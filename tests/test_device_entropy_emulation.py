"""Device entropy decoder (csrc/huff_sync_core.hpp: self-synchronising chunk decoder with speculative emission; restart segments
in chunk slots of their own) run on the CPU by tests/emu against the host front-end: same coefficients for every stream the planner declares eligible; damaged streams either stay with the host
(not eligible), raise the status flag, or decode to exactly what the host decodes."""
import ctypes as C
import io
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402
import jpeg_decoder_amd as J
import refimages as R
import synth

N = J._native


def _host(data):
    d = J.Decoder(data, device=-1)
    return d.decode_coefficients()


def _device(data):
    """-> (status, desc, [coefficient planes]) or None if the planner keeps the stream on the host."""
    L = emu.lib()
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    desc = N.ImageDesc()
    ns, nseg = C.c_uint32(0), C.c_uint32(0)
    if L.emu_huff_plan(buf, len(data), C.byref(desc), C.byref(ns), C.byref(nseg)) != 0:
        return None
    # the expansion writes every block of a scan whole, so the planes of a stream whose scans cover their planes may start as
    # anything — a pattern here; other planes start as zeros (batch.cpp's fill)
    fill = 0x5A5A if L.emu_huff_covered(buf, len(data)) == 1 else 0
    planes = [np.full(desc.components[c].block_width * desc.components[c].block_height * 64, fill, np.int16) for c in range(desc.ncomp)]
    ptrs = (C.c_void_p * 4)(*([p.ctypes.data for p in planes] + [None] * (4 - len(planes))))
    npass = C.c_uint32(0)
    st = L.emu_huff_decode(buf, len(data), ptrs, C.byref(npass))
    _device.last_passes = npass.value
    return st, desc, planes, ns.value, nseg.value


@pytest.fixture(params=[(1, 8, 2), (1, 3, 2), (1, 2, 2), (1, 3, 1), (1, 8, 1000)],
                ids=["emission", "emission-tail3", "emission-tail2", "emission-tail3-stores-one-by-one", "emission-rounds-of-eight-only"])
def emission(request):
    """Entries emitted by the sync passes and expanded into whole blocks (HuffSyncJob::emit): how much of its chunk a lane walks in
    the first sync pass (eighths: HuffSyncJob::pass0_skip), and from which pass on a lane stores its entries one by one
    (HuffSyncJob::late_pass).  (Rounds 1-3 also had a write pass after the sync passes; deleted in round 4.)"""
    emu.lib().emu_huff_set_tail(request.param[1])
    emu.lib().emu_huff_set_late(request.param[2])
    yield request.param
    emu.lib().emu_huff_set_tail(8)
    emu.lib().emu_huff_set_late(2)


def _range_by_product():
    """(max |DC * q|, max |AC * q|) the expansion of the last _device() call folded (csrc/range_stats.hpp)."""
    out = (C.c_uint32 * 2)()
    emu.lib().emu_huff_last_range(out)
    return int(out[0]), int(out[1])


def _exact_ranges(desc, planes):
    """The same two maxima, and the exact class of include/jpgpu.h, from the finished planes."""
    max_dc = max_ac = 0
    cls = 3
    for c in range(desc.ncomp):
        q = np.array(list(desc.quantization_tables[c]), np.int64).reshape(8, 8)
        s = np.abs(planes[c].astype(np.int64).reshape(-1, 8, 8) * q)
        max_dc = max(max_dc, int(s[:, 0, 0].max()))
        ac = s.copy()
        ac[:, 0, 0] = 0
        max_ac = max(max_ac, int(ac.max()))
        cls = min(cls, N.lib().jpgpu_range_class(planes[c].ctypes.data, planes[c].size, np.array(list(desc.quantization_tables[c]), np.uint16).ctypes.data))
    return max_dc, max_ac, cls


def _class_from_by_product(max_dc, max_ac):
    """range_class_from_stats without exact column sums (csrc/range_stats.hpp)."""
    if max(max_dc, max_ac) >= 1 << 15:
        return 0
    return 3 if max(max_dc + 7 * max_ac, 8 * max_ac) <= 5900 else 1


def _check_range_by_product(desc, planes):
    got = _range_by_product()
    max_dc, max_ac, exact_cls = _exact_ranges(desc, planes)
    assert got == (max_dc, max_ac), (got, max_dc, max_ac)  # the writer sees every non-zero coefficient exactly once
    assert _class_from_by_product(*got) <= exact_cls       # and the class drawn from the two maxima never overstates
    return _class_from_by_product(*got), exact_cls


def _pil_jpeg(w, h, subsampling, restart_blocks=0, restart_rows=0, gray=False, quality=85, seed=1):
    from PIL import Image
    rgb = synth.synthetic_rgb(w, h, seed=seed)
    im = Image.fromarray(rgb[..., 0] if gray else rgb)
    buf = io.BytesIO()
    kw = {"restart_marker_blocks": restart_blocks} if restart_blocks else {"restart_marker_rows": restart_rows}
    im.save(buf, format="JPEG", quality=quality, subsampling=subsampling, **kw)
    return buf.getvalue()


FIXTURES = ["reftest/restarts.jpg", "reftest/mjpeg.jpg"]


@pytest.mark.parametrize("rel", FIXTURES)
def test_reference_fixtures_with_restart_markers(rel, emission):
    data = open(os.path.join(R.GOLDEN, rel), "rb").read()
    got = _device(data)
    assert got is not None, "planner refused a plain DRI stream"
    st, desc, planes, n_scans, n_seg = got
    assert st == 0 and n_seg > n_scans >= 1
    hdesc, hcoefs = _host(data)
    assert desc.ncomp == hdesc.ncomp
    for c in range(desc.ncomp):
        assert np.array_equal(planes[c], hcoefs[c]), c
        assert list(desc.quantization_tables[c]) == list(hdesc.quantization_tables[c])
    _check_range_by_product(desc, planes)


@pytest.mark.parametrize("case", [(64, 48, "4:2:0", 0, 1), (250, 130, "4:2:0", 3, 0), (129, 257, "4:2:2", 0, 2), (200, 120, "4:4:4", 1, 0),
                                  (33, 17, "4:2:0", 5, 0), (300, 200, None, 0, 1), (1920, 64, "4:2:0", 0, 1), (17, 1080, "4:4:4", 7, 0),
                                  # a restart interval that covers the whole scan: ONE segment, decoded as a scan without markers (round 4)
                                  (96, 64, "4:2:0", 0, 50), (40, 24, "4:4:4", 500, 0)],
                         ids=lambda c: f"{c[0]}x{c[1]}-{c[2]}-b{c[3]}r{c[4]}")
def test_encoder_written_restart_streams(case, emission):
    pytest.importorskip("PIL")
    w, h, sub, rb, rr = case
    data = _pil_jpeg(w, h, sub or "4:4:4", rb, rr, gray=sub is None)
    got = _device(data)
    assert got is not None
    st, desc, planes, _ns, n_seg = got
    assert st == 0 and n_seg >= 1
    hdesc, hcoefs = _host(data)
    for c in range(desc.ncomp):
        assert np.array_equal(planes[c], hcoefs[c]), c
    got_cls, exact_cls = _check_range_by_product(desc, planes)
    assert got_cls == exact_cls == 3, (got_cls, exact_cls)  # the bench's synthetic image at quality 85: the tight class either way


def test_only_plain_sequential_streams_are_planned():
    for rel in ["benches/tower_progressive.jpg", "reftest/mozilla/jpg-progressive.jpg", "reftest/non-interleaved-mcu.jpg",
                "reftest/progressive3.jpg"]:  # progressive (the third one with DRI)
        assert _device(open(os.path.join(R.GOLDEN, rel), "rb").read()) is None, rel
    assert _device(b"") is None and _device(b"\xff\xd8\xff\xd9") is None
    # more blocks per MCU than the chunk decoder's per-block tables hold (4x4 + 1x1 + 1x1 = 18; the standard allows 10)
    pytest.importorskip("PIL")
    data = bytearray(_pil_jpeg(64, 64, "4:4:4", 0, 0))
    sof = data.find(b"\xff\xc0")
    assert sof > 0 and data[sof + 9] == 3
    assert _device(bytes(data)) is not None
    data[sof + 11] = 0x44
    assert _device(bytes(data)) is None


NO_RST = ["benches/tower.jpg", "benches/tower_grayscale.jpg", "reftest/rgb.jpg", "reftest/mozilla/jpg-gray.jpg", "reftest/mozilla/jpg-size-33x33.jpg",
          "reftest/mozilla/jpg-size-1x1.jpg", "reftest/mozilla/jpg-size-8x8.jpg", "reftest/mozilla/jpg-cmyk-1.jpg", "reftest/mozilla/jpg-cmyk-2.jpg",
          "reftest/16bit-qtables.jpg", "reftest/extraneous-data.jpg", "reftest/blank_800x280.jpg", "benches/large_image.jpg"]


@pytest.fixture(params=[(1, 256, 0), (2, 256, 0), (3, 4, 0), (2, 8, 1), (1, 256, 3)], ids=lambda p: f"iters{p[0]}-wg{p[1]}-stale{p[2]}")
def launch_shape(request):
    """Sync launches as huff.hip runs them (iterations per launch, lanes per workgroup); tiny workgroups put many
    workgroup borders — where a lane may see what an earlier batch left in the state arrays — into small streams.
    stale: the arrays start with plausible-looking states of "an earlier batch" instead of a garbage pattern."""
    emu.lib().emu_huff_set_launch(*request.param)
    yield request.param
    emu.lib().emu_huff_set_launch(1, 256, 0)


@pytest.mark.parametrize("rel", NO_RST)
def test_streams_without_restart_markers_self_synchronising_decoder(rel, launch_shape, emission):
    """Baseline files as encoders write them by default (no DRI): chunked decoding with state hand-over until the
    segmentation settles, block numbering, expansion, DC accumulation — all device code, run on the CPU."""
    data = open(os.path.join(R.GOLDEN, rel), "rb").read()
    got = _device(data)
    if got is None:
        pytest.skip("planner keeps this stream on the host (multi-scan or otherwise unusual)")
    st, desc, planes, _ns, n_seg = got
    hdesc, hcoefs = _host(data)
    assert st == 0, (st, _device.last_passes)
    assert _device.last_passes <= 6, _device.last_passes  # launches until one changed nothing
    for c in range(desc.ncomp):
        assert np.array_equal(planes[c], hcoefs[c]), c
    _check_range_by_product(desc, planes)  # (uniform scans — rgb.jpg, the CMYK files — range their DC values in the DC-sum step)


@pytest.mark.parametrize("maps", [({0: 2, 1: 3}, {0: 3, 1: 2}), ({1: 3}, {0: 2}), ({0: 3}, {1: 2}), ({}, {1: 3})], ids=lambda m: f"dc{m[0]}-ac{m[1]}".replace(" ", ""))
def test_huffman_table_ids_above_one(maps, emission):
    """Extended sequential frames may keep their tables under ids 2 and 3 (baseline: 0 and 1 only); the device tables sit in slots
    2 * id + class (huff_table_slot) and the sync pass kernel has a four-slot build for the common case."""
    pytest.importorskip("PIL")
    import jpegedit
    for (w, h, sub, rows) in [(250, 130, "4:2:0", 0), (200, 120, "4:4:4", 0), (320, 240, "4:2:0", 1)]:
        base = _pil_jpeg(w, h, sub, restart_rows=rows)
        data = jpegedit.retarget_huffman_tables(base, *maps)
        got = _device(data)
        assert got is not None
        st, desc, planes, _ns, _nseg = got
        assert st == 0
        hdesc, hcoefs = _host(data)
        bdesc, bcoefs = _host(base)
        for c in range(desc.ncomp):
            assert np.array_equal(planes[c], hcoefs[c]) and np.array_equal(planes[c], bcoefs[c]), c
        _check_range_by_product(desc, planes)


@pytest.mark.parametrize("case", [(64, 48, "4:2:0"), (250, 130, "4:2:0"), (129, 257, "4:2:2"), (200, 120, "4:4:4"), (300, 200, None),
                                  (1920, 1080, "4:2:0"), (1, 1, "4:2:0"), (17, 3000, "4:4:4")], ids=lambda c: f"{c[0]}x{c[1]}-{c[2]}")
def test_encoder_written_streams_without_restart_markers(case, launch_shape, emission):
    pytest.importorskip("PIL")
    import io
    from PIL import Image
    w, h, sub = case
    rgb = synth.synthetic_rgb(w, h, seed=w + h)
    buf = io.BytesIO()
    Image.fromarray(rgb[..., 0] if sub is None else rgb).save(buf, format="JPEG", quality=85, subsampling=sub or "4:4:4")
    data = buf.getvalue()
    got = _device(data)
    assert got is not None
    st, desc, planes, _ns, _nseg = got
    assert st == 0
    hdesc, hcoefs = _host(data)
    for c in range(desc.ncomp):
        assert np.array_equal(planes[c], hcoefs[c]), c
    assert _device.last_passes <= 8, _device.last_passes


def test_damaged_restart_streams_never_disagree_silently(emission):
    """Mutations inside the entropy data of DRI streams: not eligible, or flagged, or identical to the host."""
    pytest.importorskip("PIL")
    seeds = [open(os.path.join(R.GOLDEN, "reftest/restarts.jpg"), "rb").read(), _pil_jpeg(96, 64, "4:2:0", 2, 0), _pil_jpeg(320, 240, "4:2:0", 0, 1),
             _pil_jpeg(200, 150, "4:4:4", 0, 2, gray=True),
             open(os.path.join(R.GOLDEN, "reftest/mozilla/jpg-size-33x33.jpg"), "rb").read(), open(os.path.join(R.GOLDEN, "benches/tower.jpg"), "rb").read()]
    rng = np.random.default_rng(9)
    outcomes = {"host": 0, "flag": 0, "same": 0}
    for base in seeds:
        sos = base.rfind(b"\xff\xda")
        for trial in range(150):
            data = bytearray(base)
            for _ in range(int(rng.integers(1, 3))):
                pos = int(rng.integers(sos + 12, len(data) - 2))
                mode = int(rng.integers(0, 4))
                if mode == 0:
                    data[pos] ^= 1 << int(rng.integers(0, 8))
                elif mode == 1:
                    data[pos] = int(rng.integers(0, 256))
                elif mode == 2:
                    del data[pos]
                else:
                    data.insert(pos, int(rng.integers(0, 255)))
            data = bytes(data)
            got = _device(data)
            if got is None:
                outcomes["host"] += 1
                continue
            st, desc, planes, _a, _b = got
            if st:
                outcomes["flag"] += 1
                continue
            try:
                hdesc, hcoefs = _host(data)
            except J.Error:
                raise AssertionError("device decoder accepted a stream the host rejects")
            for c in range(desc.ncomp):
                assert np.array_equal(planes[c], hcoefs[c]), (trial, c)
            outcomes["same"] += 1
    assert outcomes["same"] > 20 and outcomes["flag"] + outcomes["host"] > 20, outcomes


def test_staging_copy_removes_exactly_the_stuffing_zeros():
    """huff_stage_segment (memchr/memcpy runs): 0xFF 0x00 -> 0xFF, everything else verbatim — 0xFF at the very end, 0xFF
    followed by something else (cannot occur inside a segment, must not be touched), runs of 0xFF 0x00, empty input; the
    slot is zero padded to huff_slot_bytes."""
    L = emu.lib()
    rng = np.random.default_rng(4)
    cases = [b"", b"\xff", b"\xff\x00", b"\x00\xff", b"\xff\x00\xff\x00\xff\x00", b"\xff\xff\x00", b"\xff\x01\xff\x00\x00", b"a" * 1000]
    for density in (0.5, 0.05, 0.004):
        for n in (1, 15, 16, 17, 257, 4096, 70001):
            a = rng.integers(0, 256, n, dtype=np.uint8)
            a[rng.random(n) < density] = 0xFF
            idx = np.flatnonzero(a[:-1] == 0xFF)
            a[idx[rng.random(idx.size) < 0.8] + 1] = 0
            cases.append(a.tobytes())
    for src in cases:
        want = bytearray()
        i = 0
        while i < len(src):
            want.append(src[i])
            if src[i] == 0xFF and i + 1 < len(src) and src[i + 1] == 0:
                i += 1
            i += 1
        slot = L.emu_slot_bytes(len(src))
        assert slot >= len(src) + 144 and slot % 16 == 0
        dst = (C.c_uint8 * (slot + 16))(*([0xAA] * (slot + 16)))
        buf = (C.c_uint8 * max(len(src), 1)).from_buffer_copy(src or b"\0")
        got = L.emu_stage_segment(dst, buf, len(src))
        out = bytes(dst)
        assert got == len(want) and out[:got] == bytes(want)
        assert out[got:slot] == bytes(slot - got) and out[slot:] == b"\xaa" * 16
        # `clean`: every 0xFF inside is followed by its stuffing zero (what the planner's short way for streams without restart
        # markers relies on: markers, fill bytes or a dangling 0xFF in the scan data send the image to the host decoder)
        want_clean = all(src[k + 1] == 0 if k + 1 < len(src) else False for k in range(len(src)) if src[k] == 0xFF)
        assert bool(L.emu_stage_segment_clean(dst, buf, len(src))) == want_clean


def test_chunk_size_follows_the_bits_per_block():
    L = emu.lib()
    assert L.emu_chunk_shift(1000, 10 ** 6) == 10          # tiny blocks: the minimum, 1,024 bits
    assert L.emu_chunk_shift(400_000, 48_960) in (11, 12)   # a 1080p 4:2:0 file of 400 kB: 65 bits per block -> 4,096 bits
    assert L.emu_chunk_shift(10 ** 7, 1000) == 15           # huge blocks: capped at 32,768 bits
    assert L.emu_chunk_shift(0, 0) == 10
    prev = 10
    for kb in range(1, 4000, 37):  # monotonic in the stream size
        sh = L.emu_chunk_shift(kb * 1000, 48_960)
        assert sh >= prev
        prev = sh


def _unary_table(symbols):
    """A Huffman table whose k-th symbol has the code 1^k 0 (lengths 1, 2, 3, ...): the shortest codes a table can have."""
    bits = [0] * 16
    for k in range(len(symbols)):
        bits[k] = 1
    return bits, list(symbols)


@pytest.mark.parametrize("kind", ["flat", "all-ones", "mixed"])
def test_emission_buffers_hold_the_densest_streams(kind, emission):
    """huff_emit_stride (csrc/huff_job.hpp): a chunk's entry list never outgrows its buffer — also when the tables give the most
    frequent symbols 1-bit codes: blocks of a DC difference of 0 and an end-of-block code (2 bits, one entry) and blocks of 63
    coefficients of +-1 (127 bits, 64 entries) are the densest a scan can be.  (The harness has a canary behind the buffers.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("baseline_encoder", os.path.join(R.ROOT if hasattr(R, "ROOT") else os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "baseline_encoder.py"))
    enc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(enc)
    import oracle as O
    w, h = 400, 296
    comps, _ = O.make_components(w, h, [(1, 1)])
    n = comps[0].block_w * comps[0].block_h
    rng = np.random.default_rng(5)
    co = np.zeros((n, 64), np.int16)
    if kind in ("all-ones", "mixed"):
        co[:, 1:] = rng.choice([-1, 1], (n, 63))
        co[:, 0] = 0
        if kind == "mixed":
            co[rng.random(n) < 0.5, 1:] = 0
    dc = _unary_table(range(12))
    ac_syms = [0x01, 0x00] if kind == "all-ones" else [0x00, 0x01]  # the 1-bit code goes to what the stream is made of
    ac = _unary_table(ac_syms + [0xF0, 0x11, 0x02, 0x21, 0x12, 0x31])
    class Cm:  # (the encoder's view of a component)
        pass
    cm = Cm()
    cm.horizontal_sampling_factor = cm.vertical_sampling_factor = 1
    cm.block_width, cm.block_height = comps[0].block_w, comps[0].block_h
    data = enc.encode_from_coefficients([cm], [np.ones(64, np.int64)], [co.reshape(-1)], w, h, huffman={"dc": [dc, dc], "ac": [ac, ac]})
    got = _device(data)
    assert got is not None
    st, desc, planes, _ns, _nseg = got
    assert (st & 0xC000) == 0, hex(st)  # never past a buffer (canary), never an entry for the wrong component
    if kind != "mixed":
        assert st == 0, st
    if st:  # (1-bit codes re-synchronise badly: such a scan may not settle in the launches it is given — flagged, the host's then)
        return
    hdesc, hcoefs = _host(data)
    assert np.array_equal(planes[0], hcoefs[0])


@pytest.mark.parametrize("case", [(320, 200, "4:2:0", 95, False), (257, 129, "4:4:4", 60, False), (400, 300, "4:2:2", 98, False), (200, 200, None, 90, False),
                                  (320, 240, "4:2:0", 92, True)], ids=lambda c: f"{c[0]}x{c[1]}-{c[2]}-q{c[3]}{'-dri' if c[4] else ''}")
def test_encoder_optimised_tables_take_the_second_level_tables(case, emission):
    """Tables an encoder fits to the image (Pillow's optimize=True) have other code lengths than Annex K's — long codes for rare
    symbols, whose prefixes get second-level tables (DevHuffTable::lut2, built from the reference's maxcode walk) or fall back to
    the walk: every symbol class must come out as the host decoder's."""
    pytest.importorskip("PIL")
    import io
    from PIL import Image
    w, h, sub, q, dri = case
    rgb = synth.synthetic_rgb(w, h, seed=w * 3 + h)
    buf = io.BytesIO()
    kw = {"restart_marker_rows": 1} if dri else {}
    Image.fromarray(rgb[..., 0] if sub is None else rgb).save(buf, format="JPEG", quality=q, optimize=True, subsampling=sub or "4:4:4", **kw)
    data = buf.getvalue()
    got = _device(data)
    assert got is not None
    st, desc, planes, _ns, _nseg = got
    assert st == 0, st
    hdesc, hcoefs = _host(data)
    for c in range(desc.ncomp):
        assert np.array_equal(planes[c], hcoefs[c]), c


def test_more_long_code_prefixes_than_second_level_tables(emission):
    """41 symbols with 11-bit codes are 21 prefixes of the 10-bit lookahead: twelve get second-level tables (HUFF_SUB_TABLES), the
    rest must take the maxcode walk — both in one stream, against the host decoder."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("baseline_encoder", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "baseline_encoder.py"))
    enc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(enc)
    import oracle as O
    w, h = 256, 192
    comps, _ = O.make_components(w, h, [(1, 1)])
    n = comps[0].block_w * comps[0].block_h
    rng = np.random.default_rng(11)
    co = np.zeros((n, 64), np.int64)
    for b in range(n):  # a few coefficients per block at random zig-zag positions: runs 0..15 (and beyond: ZRL), sizes 1..3
        pos = np.sort(rng.choice(np.arange(1, 64), int(rng.integers(0, 9)), replace=False))
        nat = enc.UNZIGZAG[pos]
        co[b, nat] = rng.integers(1, 8, pos.size) * rng.choice([-1, 1], pos.size)
    co[:, 0] = rng.integers(-60, 61, n)
    syms = [(r << 4) | s for s in (1, 2, 3) for r in range(16)]
    order = [0x00, 0x01, 0x02, 0x11, 0x03, 0x21, 0x12, 0x31, 0x41] + [s for s in syms if s not in (0x01, 0x02, 0x11, 0x03, 0x21, 0x12, 0x31, 0x41)] + [0xF0]
    bits = [0] * 16
    for ln in range(2, 11):
        bits[ln - 1] = 1
    bits[10] = len(order) - 9  # everything else: 11 bits
    ac = (bits, order)
    dc_bits = [0] * 16
    for k in range(12):
        dc_bits[k + 1] = 1  # lengths 2..13
    dc = (dc_bits, list(range(12)))
    class Cm:
        pass
    cm = Cm()
    cm.horizontal_sampling_factor = cm.vertical_sampling_factor = 1
    cm.block_width, cm.block_height = comps[0].block_w, comps[0].block_h
    data = enc.encode_from_coefficients([cm], [np.ones(64, np.int64)], [co.astype(np.int16).reshape(-1)], w, h, huffman={"dc": [dc, dc], "ac": [ac, ac]})
    got = _device(data)
    assert got is not None
    st, desc, planes, _ns, _nseg = got
    hdesc, hcoefs = _host(data)
    assert st == 0, hex(st)
    assert np.array_equal(planes[0], hcoefs[0])


@pytest.mark.parametrize("case", [(160, 120, {"restart_marker_rows": 1}), (97, 61, {"restart_marker_blocks": 7}), (320, 200, {"restart_marker_rows": 2})],
                         ids=lambda c: f"{c[0]}x{c[1]}-" + "-".join(f"{k[15:]}{v}" for k, v in c[2].items()))
def test_restart_streams_whose_components_share_their_tables(case, emission):
    """CMYK as Pillow writes it: four components, one pair of Huffman tables — a `uniform` scan (the chunk decoder cannot tell
    the blocks of an MCU apart, DC values are summed per plane afterwards) — with restart markers: the sums start again at
    every segment (huff_dc_prefix_kernel's restart intervals)."""
    pytest.importorskip("PIL")
    import io
    from PIL import Image
    w, h, kw = case
    rgb = synth.synthetic_rgb(w, h, seed=w + 7 * h)
    buf = io.BytesIO()
    Image.fromarray(np.concatenate([rgb, rgb[..., :1]], axis=2), mode="CMYK").save(buf, format="JPEG", quality=88, **kw)
    data = buf.getvalue()
    got = _device(data)
    assert got is not None
    st, desc, planes, _ns, n_seg = got
    assert st == 0 and n_seg > 1 and desc.ncomp == 4
    hdesc, hcoefs = _host(data)
    for c in range(desc.ncomp):
        assert np.array_equal(planes[c], hcoefs[c]), c
    _check_range_by_product(desc, planes)


# ---- "host light": the staging pass on the device (csrc/huff_unstuff_core.hpp) against the host's huff_stage_segment ---------------------
def _unstuff_both(raw, lead):
    """-> (device rule: bytes or None if refused, host: bytes, host says clean)"""
    L = emu.lib()
    n = len(raw)
    buf = np.zeros(((lead + n + 15) // 16 + 2) * 16, np.uint8)
    buf[:] = 0xA5  # (whatever surrounds the scan in the mirror: other scans, headers — must not matter)
    buf[lead:lead + n] = np.frombuffer(raw, np.uint8)
    dst = np.zeros(n + 16, np.uint8)
    got = L.emu_unstuff(buf.ctypes.data, lead, n, dst.ctypes.data)
    slot = np.zeros(L.emu_slot_bytes(n) + 64, np.uint8)
    src = np.frombuffer(raw, np.uint8).copy() if n else np.zeros(1, np.uint8)
    want_n = L.emu_stage_segment(slot.ctypes.data, src.ctypes.data, n)
    clean = L.emu_stage_segment_clean(slot.ctypes.data, src.ctypes.data, n) == 1
    return (None if got < 0 else dst[:got].tobytes()), slot[:want_n].tobytes(), clean


def test_device_staging_pass_equals_the_hosts():
    """Random entropy-coded-looking data (a 0xFF every ~200 bytes, each with its stuffing zero) at every alignment of the scan inside its
    16-byte pieces: same bytes as huff_stage_segment; and the rule refuses exactly what the host's pass calls unclean — a marker, a fill
    byte, a 0xFF as the scan's last byte ("a 0xFF without stuffing and a marker inside a chunk")."""
    rng = np.random.default_rng(4242)
    for trial in range(300):
        n = int(rng.integers(0, 700)) if trial % 3 else int(rng.integers(4000, 9000))
        raw = bytearray(rng.integers(0, 255, n, dtype=np.uint8).tobytes())  # (no 0xFF yet)
        i = 0
        while i + 1 < n:  # stuffed pairs
            i += int(rng.integers(1, 400))
            if i + 1 < n:
                raw[i], raw[i + 1] = 0xFF, 0x00
                i += 2
        damage = trial % 5
        if damage == 1 and n > 4:
            pos = int(rng.integers(0, n - 1))
            raw[pos], raw[pos + 1] = 0xFF, int(rng.choice([0xD0, 0xD9, 0xFF, 0x01, 0xC4]))
        elif damage == 2 and n > 0:
            raw[n - 1] = 0xFF  # nothing behind it inside the scan
        elif damage == 3 and n > 40:
            for pos in (15, 16, 31, 32):  # pairs across piece boundaries (for some alignment)
                raw[pos], raw[pos + 1] = 0xFF, 0x00
        for lead in (0, 1, 7, 15) if trial % 4 else range(16):
            got, want, clean = _unstuff_both(bytes(raw), lead)
            if clean:
                assert got == want, (trial, lead, n)
            else:
                assert got is None, (trial, lead, n)
    # the cases by hand
    assert _unstuff_both(b"\x12\xff\x00\x00\x34", 3)[0] == b"\x12\xff\x00\x34"   # the zero behind a stuffing zero is data
    assert _unstuff_both(b"\xff\x00" * 40, 9)[0] == b"\xff" * 40
    assert _unstuff_both(b"\x01\x02\xff", 0)[0] is None and _unstuff_both(b"\xff\xd9", 5)[0] is None and _unstuff_both(b"\xff\xff\x00", 0)[0] is None
    assert _unstuff_both(b"", 4)[0] == b""


@pytest.mark.parametrize("case", [(64, 48, 0, 0), (250, 130, 0, 0), (700, 260, 0, 0), (1920, 128, 0, 0), (320, 240, 0, 1), (320, 240, 0, 3), (320, 240, 7, 0),
                                  (700, 260, 5, 0), (100, 75, 1, 0), (1920, 64, 0, 1), (1920, 96, 50, 0)],
                         ids=lambda c: f"{c[0]}x{c[1]}-b{c[2]}r{c[3]}")
@pytest.mark.parametrize("quality", [85, 30, 97])
def test_entry_list_walk_reads_the_lists_as_the_expansion_does(case, quality, emission):
    """csrc/fused_entries.hpp, round 6: the 4:2:0 pixel walk reads the settled entry lists itself.  tests/emu holds a twin of its
    reading (strip index + scatter, one entry after the other, strips of 42 / 3 / 1 MCUs) that emu_huff_decode runs on every eligible
    scan next to the expansion's twin: status bit 13 if a single coefficient differs.  Restart segments: a segment's first chunk
    continues nothing, block numbers are clamped to the segment's end (the bench's restart leg found the version without the clamp:
    every image went back to the host), a chunk writes nothing beyond its segment."""
    pytest.importorskip("PIL")
    w, h, rb, rr = case
    emu.lib().emu_huff_entry_walk_checked.restype = C.c_uint32
    emu.lib().emu_huff_entry_walk_checked()
    data = _pil_jpeg(w, h, "4:2:0", rb, rr, quality=quality, seed=w + quality) if (rb or rr) else _pil_jpeg_plain(w, h, quality, seed=w + quality)
    got = _device(data)
    assert got is not None
    st, desc, planes, _ns, _nseg = got
    assert st == 0, hex(st)
    assert emu.lib().emu_huff_entry_walk_checked() == 1  # (the twin did read this scan)
    _hdesc, hcoefs = _host(data)
    for c in range(desc.ncomp):
        assert np.array_equal(planes[c], hcoefs[c]), c


def _pil_jpeg_plain(w, h, quality, seed):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(synth.synthetic_rgb(w, h, seed=seed)).save(buf, format="JPEG", quality=quality, subsampling="4:2:0")
    return buf.getvalue()


def test_entry_list_walk_twin_sees_the_reference_fixtures():
    n = 0
    emu.lib().emu_huff_entry_walk_checked.restype = C.c_uint32
    emu.lib().emu_huff_entry_walk_checked()
    import glob
    for name in sorted(glob.glob(os.path.join(R.GOLDEN, "reftest", "**", "*.jp*g"), recursive=True)):
        data = open(name, "rb").read()
        try:
            got = _device(data)
        except Exception:
            continue
        if got is None:
            continue
        st = got[0]
        assert not (st & 0x2000), name  # (whatever else the stream's status says, the two readings of its lists agree)
        n += emu.lib().emu_huff_entry_walk_checked()
    assert n >= 3, n
