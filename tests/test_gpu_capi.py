"""The drop-in boundary exercised from C, not through ctypes: tests/capi/worker_roundtrip.c performs exactly the call
sequence of the Rust shim (rust/src/worker/hip.rs) — create / start / append_row x MCU rows / finish_plane / compute_image —
against libjpgpu.so, and the pixels are compared with the oracle here.  The statuses the shim maps onto `Error`
(include/jpgpu.h:28-36 <-> src/error.rs:36-48) are checked on the reference's own error cases."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

import jpeg_decoder_amd as J
import oracle as O
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "capi", "worker_roundtrip.c")
CT = dict(NONE=0, UNKNOWN=1, GRAYSCALE=2, RGB=3, YCBCR=4, CMYK=5, YCCK=6, JCS_BG_YCC=7, JCS_BG_RGB=8)


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    J.build()
    out = str(tmp_path_factory.mktemp("capi") / "worker_roundtrip")
    libdir = os.path.join(ROOT, "jpeg-decoder_amd")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Wextra", "-std=c11", "-I", os.path.join(ROOT, "include"), SRC, "-o", out,
                           "-L", libdir, "-ljpgpu", "-Wl,-rpath," + libdir])
    return out


def _case_file(path, ocomps, qts, coefs, w_, h_, ct, resident, short_rows=0):
    with open(path, "wb") as f:
        f.write(struct.pack("<IIIiI", len(ocomps), w_, h_, CT[ct], 1 if resident else 0))
        for i, oc in enumerate(ocomps):
            jc = J.Component(oc.identifier, oc.h, oc.v, oc.tq, oc.dct_scale, oc.size_w, oc.size_h, oc.block_w, oc.block_h)
            f.write(bytes(jc))
            f.write(np.ascontiguousarray(qts[i], dtype=np.uint16).tobytes())
            per_row = oc.block_w * oc.v * 64
            n_rows = len(coefs[i]) // per_row - short_rows
            f.write(struct.pack("<I", n_rows))
            f.write(np.ascontiguousarray(coefs[i][: n_rows * per_row], dtype=np.int16).tobytes())


def _run(exe, tmp_path, *args, **kw):
    case, out = str(tmp_path / "case.bin"), str(tmp_path / "pixels.bin")
    _case_file(case, *args, **kw)
    r = subprocess.run([exe, case, out], capture_output=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    blob = open(out, "rb").read()
    return struct.unpack("<i", blob[:4])[0], blob[4:]


CASES = [(160, 96, [(2, 2), (1, 1), (1, 1)], "YCBCR"), (45, 29, [(1, 1), (1, 1), (1, 1)], "YCBCR"), (64, 24, [(2, 1), (1, 1), (1, 1)], "YCBCR"),
         (37, 21, [(1, 1)], "GRAYSCALE"), (50, 61, [(1, 2), (1, 1), (1, 1)], "YCBCR"), (40, 40, [(1, 1)] * 4, "CMYK"), (33, 17, [(1, 1)] * 3, "RGB")]


@pytest.mark.parametrize("resident", [True, False], ids=["device-resident", "compat"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[1]}-{len(c[2])}c{c[2][0][0]}{c[2][0][1]}-{c[3]}")
def test_shim_call_sequence_from_c_matches_the_oracle(exe, tmp_path, case, resident):
    w_, h_, samp, ct = case
    rng = np.random.default_rng(w_ * 1000 + h_)
    ocomps, _ = O.make_components(w_, h_, samp)
    qts = [rng.integers(1, 200, 64).astype(np.uint16) for _ in ocomps]
    coefs = [synth.sparse_coefficients(rng, c.block_w * c.block_h) for c in ocomps]
    status, payload = _run(exe, tmp_path, ocomps, qts, coefs, w_, h_, ct, resident)
    assert status == 0, payload
    want = O.pixels_from_coefficients(ocomps, qts, coefs, w_, h_, ct)
    assert np.array_equal(np.frombuffer(payload, np.uint8), want)


def test_partial_planes_are_zero_below_the_appended_rows(exe, tmp_path):
    """A scan that ends early (fewer append_row calls than the plane has MCU rows): the rest of the plane is the zeroed
    Vec of src/worker/immediate.rs:34-36."""
    w_, h_, samp = 64, 64, [(1, 1), (1, 1), (1, 1)]
    rng = np.random.default_rng(5)
    ocomps, _ = O.make_components(w_, h_, samp)
    qts = [rng.integers(1, 64, 64).astype(np.uint16) for _ in ocomps]
    coefs = [synth.sparse_coefficients(rng, c.block_w * c.block_h) for c in ocomps]
    status, payload = _run(exe, tmp_path, ocomps, qts, coefs, w_, h_, "YCBCR", True, short_rows=3)
    assert status == 0, payload
    planes = [O.idct_plane(ocomps[i], qts[i], coefs[i], n_mcu_rows=ocomps[i].block_h - 3) for i in range(3)]
    assert np.array_equal(np.frombuffer(payload, np.uint8), O.compute_image(ocomps, planes, w_, h_, "YCBCR"))


def test_statuses_the_shim_maps_onto_error(exe, tmp_path):
    rng = np.random.default_rng(9)
    # (3,1)(2,1): Upsampler::new refuses the ratio (src/upsampler.rs:97-99) -> Error::Unsupported(NonIntegerSubsamplingRatio) = status 2
    ocomps, _ = O.make_components(48, 16, [(3, 1), (2, 1), (1, 1)])
    qts = [np.ones(64, np.uint16) for _ in ocomps]
    coefs = [np.zeros(c.block_w * c.block_h * 64, np.int16) for c in ocomps]
    status, payload = _run(exe, tmp_path, ocomps, qts, coefs, 48, 16, "YCBCR", True)
    assert status == 2 and b"NonIntegerSubsamplingRatio" in payload
    # a colour transform compute_image does not implement (src/decoder.rs:1391-1399) -> Error::Unsupported(ColorTransform(..))
    ocomps, _ = O.make_components(16, 16, [(1, 1)] * 3)
    qts = [np.ones(64, np.uint16) for _ in ocomps]
    coefs = [np.zeros(c.block_w * c.block_h * 64, np.int16) for c in ocomps]
    status, payload = _run(exe, tmp_path, ocomps, qts, coefs, 16, 16, "JCS_BG_YCC", True)
    assert status == 2 and b"ColorTransform" in payload
    # three components under a 4-component transform: choose_color_convert_func panics in the reference -> Error::Internal = status 4
    status, payload = _run(exe, tmp_path, ocomps, qts, coefs, 16, 16, "CMYK", True)
    assert status in (1, 4), (status, payload)
