"""Device decoder for PROGRESSIVE frames (csrc/huff_prog_wave.hpp: one wave per scan, coefficients accumulated in
place; SURVEY 8f n3 / BASELINE configs[3]) run on the CPU by tests/emu against the host front-end, which restates
src/decoder.rs:1086-1298: the same coefficient planes for every stream the planner (Frontend::plan_progressive_scans) declares eligible,
whatever the order the tracks are walked in; damaged streams stay with the host (not eligible), raise the status word, or decode to
exactly what the host decodes."""
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
    return J.Decoder(data, device=-1).decode_coefficients()


def _plan(data):
    L = emu.lib()
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    desc = N.ImageDesc()
    ns, nt = C.c_uint32(0), C.c_uint32(0)
    if L.emu_prog_plan(buf, len(data), C.byref(desc), C.byref(ns), C.byref(nt)) != 0:
        return None
    return desc, ns.value, nt.value


def _device(data, order=0):
    """-> (status, desc, planes, scans, tracks) or None if the planner keeps the stream on the host."""
    pl = _plan(data)
    if pl is None:
        return None
    desc, ns, nt = pl
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data)
    planes = [np.zeros(desc.components[c].block_width * desc.components[c].block_height * 64, np.int16) for c in range(desc.ncomp)]
    ptrs = (C.c_void_p * 4)(*([p.ctypes.data for p in planes] + [None] * (4 - len(planes))))
    st = emu.lib().emu_prog_decode(buf, len(data), ptrs, order)
    return st, desc, planes, ns, nt


def _pil(w, h, subsampling, gray=False, quality=85, seed=1, **kw):
    from PIL import Image
    rgb = synth.synthetic_rgb(w, h, seed=seed)
    im = Image.fromarray(rgb[..., 0] if gray else rgb)
    buf = io.BytesIO()
    im.save(buf, format="JPEG", quality=quality, subsampling=subsampling, progressive=True, **kw)
    return buf.getvalue()


def _same_as_host(data, order=0):
    got = _device(data, order)
    assert got is not None, "the planner refused a plain progressive stream"
    st, desc, planes, ns, nt = got
    assert st == 0, hex(st)
    hdesc, hcoefs = _host(data)
    assert desc.ncomp == hdesc.ncomp
    for c in range(desc.ncomp):
        assert np.array_equal(planes[c], np.asarray(hcoefs[c], np.int16)), c
        assert list(desc.quantization_tables[c]) == list(hdesc.quantization_tables[c])
    return ns, nt


FIXTURES = {  # file: (scans, tracks) — every progressive file of the reference's corpora that decodes (the others stay with the host)
    "benches/tower_progressive.jpg": (10, 4), "reftest/mozilla/jpg-progressive.jpg": (10, 4), "reftest/progressive3.jpg": (11, 8),
    "reftest/progressive-missing-ac.jpg": (1, 1), "reftest/progressive-missing-dc.jpg": (1, 1),
}


@pytest.mark.parametrize("rel", sorted(FIXTURES))
@pytest.mark.parametrize("order", [0, 1, 2, 3, 4, 5], ids=["tracks-in-order", "tracks-reversed", "scan-by-scan-round-robin", "a-wave-per-scan-with-its-waits",
                                                            "a-wave-per-scan-with-its-waits-again", "a-wave-per-track"])
def test_reference_fixtures(rel, order):
    data = open(os.path.join(R.GOLDEN, rel), "rb").read()
    assert _same_as_host(data, order) == FIXTURES[rel]


def test_streams_the_host_keeps():
    for rel in ("reftest/partial_progressive.jpg",    # ends inside a scan: the reference's partial render is the host's business
                "benches/tower.jpg",                  # not progressive
                "reftest/non-interleaved-mcu.jpg"):
        assert _plan(open(os.path.join(R.GOLDEN, rel), "rb").read()) is None, rel


@pytest.mark.parametrize("case", [(64, 48, "4:2:0"), (250, 130, "4:2:0"), (129, 257, "4:2:2"), (200, 120, "4:4:4"), (33, 17, "4:2:0"), (300, 200, None),
                                  (8, 8, "4:4:4"), (1, 1, "4:2:0"), (17, 9, "4:2:2"), (512, 16, "4:2:0"), (16, 512, "4:4:4"), (640, 480, "4:2:0")],
                         ids=lambda c: f"{c[0]}x{c[1]}-{c[2]}")
@pytest.mark.parametrize("quality", [35, 85, 98])
def test_encoder_written_progressive_streams(case, quality):
    """libjpeg-turbo's default progressive script (DC first, AC bands with successive approximation, optimised tables per scan) at
    several sizes and samplings: ragged edges, one-MCU images, non-interleaved AC scans of subsampled components."""
    pytest.importorskip("PIL")
    w, h, sub = case
    data = _pil(w, h, sub or "4:4:4", gray=sub is None, quality=quality, seed=w + h)
    ns, nt = _same_as_host(data, order=(w + quality) % 4)
    assert _same_as_host(data, order=4 + (w + quality) % 2) == (ns, nt)  # round 6: the wave-per-scan walk (huff_prog_wave.hpp)
    assert ns >= 3 and nt >= 2 - (sub is None)


def test_tracks_are_the_connected_bands():
    """Scans that share a coefficient of a component are one track; the DC scans of all components another; nothing else joins."""
    pytest.importorskip("PIL")
    data = _pil(96, 64, "4:2:0")
    _desc, ns, nt = _plan(data)
    assert (ns, nt) == (10, 4)  # DC first + refine | Y: 1-5, 6-63, two refinements | Cb: 1-63 + refinement | Cr likewise
    _desc, ns, nt = _plan(_pil(96, 64, "4:4:4", gray=True))
    assert nt == 2  # DC | AC


def test_scans_pipelined_over_lanes_wait_for_the_last_writer_of_their_coefficients():
    """huff_prog_job.hpp, "a wave per scan": a scan stays behind, block for block, the LAST earlier scan that covered each of its
    coefficients — tower_progressive.jpg (libjpeg's default script): DC refinement behind DC first; Y's first refinement behind BOTH
    first scans of its band (1-5 and 6-63), its second behind the first; the first scans behind nobody."""
    data = open(os.path.join(R.GOLDEN, "benches", "tower_progressive.jpg"), "rb").read()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    deps = np.full(3 * 16, -7, np.int32)
    rank = np.zeros(16, np.uint32)
    whole = np.zeros(3 * 16, np.uint32)
    n = emu.lib().emu_prog_dependencies(buf, len(data), deps.ctypes.data, rank.ctypes.data, whole.ctypes.data, 16)
    assert n == 10
    d = [sorted(int(x) for x in deps[3 * j:3 * j + 3] if x >= 0) for j in range(n)]
    # scans in stream order: 0 DC first | 1 Y 1-5 | 2 Cr 1-63 | 3 Cb 1-63 | 4 Y 6-63 | 5 Y refine 2:1 | 6 DC refine | 7 Cr refine | 8 Cb refine | 9 Y refine 1:0
    assert d == [[], [], [], [], [], [1, 4], [0], [2], [3], [5]]
    assert list(rank[:n]) == [0, 0, 0, 0, 0, 1, 1, 1, 1, 2]
    assert not whole[:3 * n].any()  # every pair walks the same blocks in the same order: block for block
    # a frame whose DC refinement is NOT interleaved like its first DC scan must wait for the whole of it
    pytest.importorskip("PIL")
    for sub in ("4:2:0", "4:4:4"):
        f = _pil(64, 48, sub)
        b2 = (C.c_uint8 * len(f)).from_buffer_copy(f)
        n2 = emu.lib().emu_prog_dependencies(b2, len(f), deps.ctypes.data, rank.ctypes.data, whole.ctypes.data, 16)
        assert n2 == 10 and max(rank[:n2]) == 2


def _damage(rng, base):
    d = bytearray(base)
    lo = max(2, len(d) // 5)
    for _ in range(int(rng.integers(1, 4))):
        pos = int(rng.integers(lo, len(d) - 2))
        mode = int(rng.integers(0, 5))
        if mode == 0:
            d[pos] ^= 1 << int(rng.integers(0, 8))
        elif mode == 1:
            del d[pos]
        elif mode == 2:
            d[pos] = 0xFF
        elif mode == 3:
            del d[pos:pos + int(rng.integers(1, 40))]
        else:
            d[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 6)), dtype=np.uint8))
    return bytes(d)


@pytest.mark.parametrize("seed", range(6))
def test_damaged_streams_never_differ_from_the_host(seed):
    """Bit flips, deletions, stray 0xFF bytes and insertions anywhere behind the first fifth of a progressive file (headers of later
    scans and their Huffman tables included): the planner refuses the stream, or the walk raises the status word (the host then decodes
    the image), or the planes are the host's — never different planes with a clean status."""
    pytest.importorskip("PIL")
    rng = np.random.default_rng(7100 + seed)
    bases = [open(os.path.join(R.GOLDEN, "benches", "tower_progressive.jpg"), "rb").read(), _pil(120, 72, "4:2:0", seed=seed), _pil(64, 64, "4:4:4", quality=60, seed=seed + 9)]
    kept = flagged = same = 0
    for t in range(60):
        data = _damage(rng, bases[t % len(bases)])
        got = _device(data, order=t % 6)
        if got is None:
            kept += 1
            continue
        st, desc, planes, _ns, _nt = got
        if st != 0:
            assert st & 1
            flagged += 1
            continue
        try:
            _hdesc, hcoefs = _host(data)
        except J.Error:
            # the host's marker loop or entropy decoder raises where the device decoded something: only acceptable if the device's
            # result is never used — the pipeline decodes such a stream on the host when the PLANNER refuses it; a clean device
            # status with a host error would be a wrong answer
            raise AssertionError("device decoded a stream the host refuses")
        for c in range(desc.ncomp):
            assert np.array_equal(planes[c], np.asarray(hcoefs[c], np.int16)), (t, c)
        same += 1
    assert kept + flagged + same == 60 and same > 0


def test_many_threads_many_table_definitions():
    """The front-end keeps ONE copy of every Huffman table definition in a process-wide registry (64 shards by a hash of the definition,
    csrc/host/frontend.cpp build_cached) and per thread pointers to the tables it used last.  Progressive files from a real encoder carry
    twelve optimised tables each, no two files alike: eight threads plan and decode 48 different frames over and over — more definitions
    than the registry and the threads' slots hold, so entries are replaced all the time — and every plan and every coefficient plane
    equals what one thread alone gets."""
    pytest.importorskip("PIL")
    import threading
    frames = [_pil(64 + 8 * (i % 5), 48 + 8 * (i % 3), ("4:2:0", "4:4:4", "4:2:2")[i % 3], quality=50 + (i * 7) % 45, seed=500 + i) for i in range(48)]
    alone = []
    for f in frames:
        st, desc, planes, ns, nt = _device(f, 0)
        assert st == 0
        alone.append((ns, nt, [p.copy() for p in planes]))
    bad = []

    def work(t):
        try:
            for rep in range(3):
                for i in range(len(frames)):
                    j = (i * 5 + t * 7 + rep) % len(frames)
                    st, desc, planes, ns, nt = _device(frames[j], (t + rep) % 4)
                    if st != 0 or (ns, nt) != alone[j][:2] or not all(np.array_equal(a, b_) for a, b_ in zip(planes, alone[j][2])):
                        bad.append((t, rep, j))
                    if (i + t) % 6 == 0:  # (the host decoder goes through the same registry)
                        _hd, hc = _host(frames[j])
                        if not all(np.array_equal(np.asarray(hc[c], np.int16), alone[j][2][c]) for c in range(len(alone[j][2]))):
                            bad.append((t, rep, j, "host"))
        except Exception as e:  # noqa: BLE001
            bad.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not bad, bad[:3]


def test_prog_table_is_the_reference_procedure():
    """ProgHuffTable = the reference's 8-bit lookup + maxcode walk (src/huffman.rs:31-58): every code of an optimised table decodes
    to its symbol through the emulated lane (covered by the stream tests) — here: the struct's size, which the LDS layout is built on."""
    assert N.lib() is not None
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jpeg-decoder_amd", "csrc", "huff_prog_job.hpp")).read()
    assert re.search(r"sizeof\(ProgHuffTable\) == 912", text)


# ---- scan scripts no Pillow writes (tools/progressive_encoder.py: what `jpegtran -scans` would make) ------------------------------------
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def _scripted(w, h, sampling, script, quality=85, seed=3):
    import progressive_encoder as P
    return P.encode_rgb(synth.synthetic_rgb(w, h, seed=seed), script, quality=quality, sampling=sampling)


@pytest.mark.parametrize("order", [0, 2, 3, 4, 5], ids=["tracks-in-order", "scan-by-scan-round-robin", "a-wave-per-scan", "a-wave-per-scan-again", "a-wave-per-track"])
def test_split_band_refinement_scripts(order):
    """a mask word covers all 63 AC positions of a block, a scan only its band — with Y 1-5 | Y 6-63 | refine 1-5 |
    refine 6-63 the refinement of one band runs beside the first scan of the other on the same blocks (the device publishes its mask
    bits with atomic ORs; here the walks run one after the other and must agree with the host whatever the order)."""
    import progressive_encoder as P
    for w, h, samp, script in ((96, 64, "444", P.SPLIT_REFINEMENT_YCC), (120, 72, "420", P.SPLIT_REFINEMENT_YCC), (77, 53, "gray", P.SPLIT_REFINEMENT_GRAY),
                               (250, 130, "422", P.SPLIT_REFINEMENT_YCC)):
        ns, nt = _same_as_host(_scripted(w, h, samp, script), order)
        assert ns == len(script) and nt >= 3


@pytest.mark.parametrize("seed", range(8))
def test_random_scan_scripts(seed):
    """Bands cut at random places, refinements in random order between the scans of other bands and components, DC interleaved or
    per component: every legal script decodes to the host's planes on every walk."""
    import progressive_encoder as P
    rng = np.random.default_rng(4200 + seed)
    for t in range(6):
        nc = 1 if (seed + t) % 4 == 0 else 3
        samp = "gray" if nc == 1 else ("444", "420", "422")[t % 3]
        script = P.random_script(rng, nc)
        data = _scripted(40 + 9 * t + seed, 33 + 7 * t, samp, script, quality=50 + 6 * t, seed=seed * 10 + t)
        for order in (0, 3, 4, 5):
            got = _device(data, order)
            if order == 3 and got is not None and got[0] == -2:  # (more than three producers for one scan: walked as tracks)
                continue
            assert got is not None, script
            st, desc, planes, ns, _nt = got
            assert st == 0 and ns == len(script), (hex(st), script)
            _hdesc, hcoefs = _host(data)
            for c in range(desc.ncomp):
                assert np.array_equal(planes[c], np.asarray(hcoefs[c], np.int16)), (order, c, script)
