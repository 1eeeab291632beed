"""Device entropy decoder (csrc/huff_core.hpp: one lane per restart segment) run on the CPU by tests/emu against the host
front-end: same coefficients for every stream the planner declares eligible; damaged streams either stay with the host
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
    planes = [np.zeros(desc.components[c].block_width * desc.components[c].block_height * 64, np.int16) for c in range(desc.ncomp)]
    ptrs = (C.c_void_p * 4)(*([p.ctypes.data for p in planes] + [None] * (4 - len(planes))))
    st = L.emu_huff_decode(buf, len(data), ptrs)
    return st, desc, planes, ns.value, nseg.value


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
def test_reference_fixtures_with_restart_markers(rel):
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


@pytest.mark.parametrize("case", [(64, 48, "4:2:0", 0, 1), (250, 130, "4:2:0", 3, 0), (129, 257, "4:2:2", 0, 2), (200, 120, "4:4:4", 1, 0),
                                  (33, 17, "4:2:0", 5, 0), (300, 200, None, 0, 1), (1920, 64, "4:2:0", 0, 1), (17, 1080, "4:4:4", 7, 0)],
                         ids=lambda c: f"{c[0]}x{c[1]}-{c[2]}-b{c[3]}r{c[4]}")
def test_encoder_written_restart_streams(case):
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


def test_streams_without_restart_markers_stay_on_the_host():
    for rel in ["benches/tower.jpg", "benches/tower_progressive.jpg", "reftest/mozilla/jpg-progressive.jpg", "reftest/rgb.jpg",
                "reftest/mozilla/jpg-gray.jpg", "reftest/non-interleaved-mcu.jpg"]:  # (the last: progressive, with DRI)
        assert _device(open(os.path.join(R.GOLDEN, rel), "rb").read()) is None, rel
    assert _device(b"") is None and _device(b"\xff\xd8\xff\xd9") is None


def test_damaged_restart_streams_never_disagree_silently():
    """Mutations inside the entropy data of DRI streams: not eligible, or flagged, or identical to the host."""
    pytest.importorskip("PIL")
    seeds = [open(os.path.join(R.GOLDEN, "reftest/restarts.jpg"), "rb").read(), _pil_jpeg(96, 64, "4:2:0", 2, 0)]
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
