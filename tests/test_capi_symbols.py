"""The C-ABI library loads on a GPU-less machine and exports every entry point that include/*.h
declares (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

import jpeg_decoder_amd as J

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in ("jpgpu.h", "jpgpu_decoder.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(jpgpu_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_boundary():
    fns = declared_functions()
    for must in ("jpgpu_worker_start", "jpgpu_worker_append_row", "jpgpu_worker_append_rows", "jpgpu_worker_get_result",
                 "jpgpu_compute_image", "jpgpu_batch_decode", "jpgpu_decoder_decode", "jpgpu_decoder_read_info"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    J.build()
    lib = C.CDLL(J._native.LIB_PATH)
    missing = [f for f in declared_functions() if not hasattr(lib, f)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    assert sorted(J._native.exported_symbols()) == declared_functions()


def test_no_device_fails_loudly_not_silently():
    if J.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(J.NoDeviceError):
        J.HipWorker()
    comps, _ = J.make_components(16, 16, [(1, 1)])
    with pytest.raises(J.Error):
        J.Batch([J.image_desc(list(comps), [[1] * 64], 16, 16, "Grayscale")])


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the package (or include/) may reference it."""
    pkg = os.path.join(ROOT, "jpeg-decoder_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dp, f), errors="replace").read()
                assert "liboracle" not in text and "import oracle" not in text and "jpeg_oracle.h" not in text, os.path.join(dp, f)


def test_range_class_host_function_matches_its_definition():
    """jpgpu_range_class (pure host code in the library): 0 hostile, 1 every |c*q| < 2^15, 3 additionally every block
    column's sum of |c*q| <= 5900 (include/jpgpu.h)."""
    import numpy as np
    lib = J.lib()
    rng = np.random.default_rng(5)

    def cls(c, q):
        c = np.ascontiguousarray(c, np.int16)
        q = np.ascontiguousarray(q, np.uint16)
        return lib.jpgpu_range_class(c.ctypes.data, c.size, q.ctypes.data)

    def want(c, q):
        s = np.abs(c.astype(np.int64).reshape(-1, 8, 8) * q.astype(np.int64).reshape(1, 8, 8))
        if s.max(initial=0) >= 1 << 15:
            return 0
        return 3 if s.sum(axis=1).max(initial=0) <= 5900 else 1

    q = rng.integers(1, 40, 64).astype(np.uint16)
    for _ in range(200):
        amp = int(rng.choice([3, 20, 200, 3000, 32767]))
        c = rng.integers(-amp, amp + 1, 64 * 7).astype(np.int16)
        assert cls(c, q) == want(c, q)
    one = np.zeros(64, np.int16); one[8] = 5900; ones = np.ones(64, np.uint16)
    assert cls(one, ones) == 3
    one[16] = 1
    assert cls(one, ones) == 1
    one[:] = 0; one[5] = 32767
    assert cls(one, ones) == 1 and cls(one, ones * 2) == 0
    assert cls(np.zeros(0, np.int16), ones) == 3


def test_pipeline_needs_a_device():
    if J.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(J.NoDeviceError):
        J.Pipeline()


def _expand_compact(buf, n_blocks):
    """Reference reading of the compact format of include/jpgpu.h (numpy, test side)."""
    import numpy as np
    bm = np.frombuffer(buf, np.uint64, n_blocks, 0)
    first = np.frombuffer(buf, np.uint32, n_blocks, 8 * n_blocks)
    values = np.frombuffer(buf, np.int16, (len(buf) - 12 * n_blocks) // 2, 12 * n_blocks)
    out = np.zeros((n_blocks, 64), np.int16)
    for b in range(n_blocks):
        ks = [k for k in range(64) if (int(bm[b]) >> k) & 1]
        out[b, ks] = values[int(first[b]): int(first[b]) + len(ks)]
    return out.reshape(-1)


def test_compact_encoder_round_trip_and_range_class():
    """jpgpu_compact_encode (pure host code): bitmap + index + values reproduce the dense blocks; size and range class
    as documented."""
    import numpy as np
    import synth
    lib = J.lib()
    rng = np.random.default_rng(11)
    q = rng.integers(1, 60, 64).astype(np.uint16)
    for nblk, maker in [(1, lambda: np.zeros(64, np.int16)), (37, lambda: synth.sparse_coefficients(rng, 37)),
                        (5, lambda: rng.integers(-32768, 32768, 5 * 64).astype(np.int16)), (0, lambda: np.zeros(0, np.int16))]:
        c = np.ascontiguousarray(maker(), np.int16)
        buf = np.zeros(lib.jpgpu_compact_max_bytes(nblk) + 16, np.uint8)
        rc = C.c_int(-1)
        n = lib.jpgpu_compact_encode(c.ctypes.data, nblk, q.ctypes.data, buf.ctypes.data, C.byref(rc))
        assert n == 12 * nblk + 2 * int(np.count_nonzero(c)) <= lib.jpgpu_compact_max_bytes(nblk)
        assert (buf[n:] == 0).all()
        assert np.array_equal(_expand_compact(buf[:n].tobytes(), nblk), c)
        assert rc.value == lib.jpgpu_range_class(c.ctypes.data, c.size, q.ctypes.data)


def test_compact_encoder_range_class_on_hostile_and_16bit_tables():
    """The SIMD block-total shortcut of the compact encoder must never promote hostile data: extreme coefficients with
    8-bit and 16-bit tables, block totals just below / above the class-3 limit."""
    import numpy as np
    lib = J.lib()
    rng = np.random.default_rng(3)

    def both(c, q):
        c = np.ascontiguousarray(c, np.int16)
        q = np.ascontiguousarray(q, np.uint16)
        buf = np.zeros(lib.jpgpu_compact_max_bytes(c.size // 64), np.uint8)
        rc = C.c_int(-1)
        lib.jpgpu_compact_encode(c.ctypes.data, c.size // 64, q.ctypes.data, buf.ctypes.data, C.byref(rc))
        return rc.value, lib.jpgpu_range_class(c.ctypes.data, c.size, q.ctypes.data)

    for q in (np.full(64, 255, np.uint16), np.full(64, 65535, np.uint16), np.full(64, 32767, np.uint16), np.ones(64, np.uint16),
              rng.integers(1, 256, 64).astype(np.uint16), rng.integers(1, 65536, 64).astype(np.uint16)):
        for c in (np.full(64, -32768, np.int16), np.full(64, 32767, np.int16), rng.integers(-32768, 32768, 640).astype(np.int16),
                  np.zeros(64, np.int16), rng.integers(-3, 4, 640).astype(np.int16), rng.integers(-40, 41, 640).astype(np.int16)):
            got, want = both(c, q)
            assert got == want, (q[:4], c[:4], got, want)
    ones = np.ones(64, np.uint16)
    edge = np.zeros(64, np.int16); edge[0:8] = [737, 737, 737, 737, 738, 738, 738, 738]   # block total 5900: every column sum <= 5900
    assert both(edge, ones) == (3, 3)
    edge[8] = 5164  # block total now above the limit, column 0 sum = 737 + 5164 = 5901 -> class 1
    assert both(edge, ones) == (1, 1)
    edge[8] = 5163  # column 0 sum exactly 5900, block total far above: the exact path must still say 3
    assert both(edge, ones) == (3, 3)


def test_library_does_not_need_the_tracing_library_to_load():
    """roctx ranges are resolved at run time (dlopen) and are no-ops where libroctx64 is absent: the library
    must not carry a load-time dependency on it."""
    import shutil
    import subprocess
    J.build()
    tool = shutil.which("readelf") or shutil.which("objdump")
    if not tool:
        pytest.skip("no readelf / objdump")
    args = [tool, "-d", J._native.LIB_PATH] if tool.endswith("readelf") else [tool, "-p", J._native.LIB_PATH]
    out = subprocess.run(args, capture_output=True, text=True).stdout
    assert "NEEDED" in out and "roctx" not in out, [l for l in out.splitlines() if "roctx" in l]
