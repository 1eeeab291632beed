"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py.  The product package never imports this module.
See oracle/jpeg_oracle.h for what is restated and how the oracle is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CT = dict(NONE=0, UNKNOWN=1, GRAYSCALE=2, RGB=3, YCBCR=4, CMYK=5, YCCK=6, JCS_BG_YCC=7, JCS_BG_RGB=8, AUTO=-1)
STATUS = {0: "Ok", 1: "Format", 2: "Unsupported", 3: "Io", 4: "Internal"}


class Component(C.Structure):
    _fields_ = [
        ("identifier", C.c_uint8),
        ("h", C.c_uint8),
        ("v", C.c_uint8),
        ("tq", C.c_uint8),
        ("dct_scale", C.c_uint32),
        ("size_w", C.c_uint16),
        ("size_h", C.c_uint16),
        ("block_w", C.c_uint16),
        ("block_h", C.c_uint16),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class _Result(C.Structure):
    _fields_ = [
        ("status", C.c_int),
        ("message", C.c_char * 160),
        ("width", C.c_uint16),
        ("height", C.c_uint16),
        ("image_w", C.c_uint16),
        ("image_h", C.c_uint16),
        ("ncomp", C.c_int),
        ("coding_process", C.c_int),
        ("is_baseline", C.c_int),
        ("color_transform", C.c_int),
        ("mcu_w", C.c_uint16),
        ("mcu_h", C.c_uint16),
        ("components", Component * 4),
        ("pixels", C.POINTER(C.c_uint8)),
        ("pixels_len", C.c_size_t),
        ("coefs", C.POINTER(C.c_int16) * 4),
        ("coefs_len", C.c_size_t * 4),
        ("qtables", (C.c_uint16 * 64) * 4),
        ("have_plane", C.c_int * 4),
        ("planes", C.POINTER(C.c_uint8) * 4),
        ("planes_len", C.c_size_t * 4),
    ]


def build(force=False):
    """Compile liboracle.so with gcc (a few hundred ms)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle_pixels.c", "oracle_front.c", "oracle_batch.c", "jpeg_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


NATIVE_CFLAGS = "-O3 -march=native -fPIC -std=c11 -fwrapv -fno-strict-aliasing"


def use_native_build():
    """bench.py's cpu_baseline leg only: rebuild the comparator ON THIS BOX with -O3 -march=native as
    oracle/liboracle_native.so and bind it in place of the portable -O2 library the tests use.  Returns the flags in
    effect (the portable ones if the native build is not possible here)."""
    global _LIB
    so = os.path.join(_HERE, "liboracle_native.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle_native.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _LIB = None
        lib(so)
        return NATIVE_CFLAGS
    except (subprocess.CalledProcessError, OSError):
        _LIB = None
        lib()
        return "-O2 (portable build; native rebuild failed)"


def lib(path=None):
    global _LIB
    if _LIB is None:
        L = C.CDLL(path or build())
        L.orc_dequantize_and_idct_block.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_dequantize_and_idct_block.restype = None
        L.orc_choose_idct_size.argtypes = [C.c_uint16] * 4
        L.orc_choose_idct_size.restype = C.c_int
        L.orc_update_component_sizes.argtypes = [C.c_uint16, C.c_uint16, C.POINTER(Component), C.c_int,
                                                 C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]
        L.orc_update_component_sizes.restype = C.c_int
        L.orc_plane_bytes.argtypes = [C.POINTER(Component)]
        L.orc_plane_bytes.restype = C.c_size_t
        L.orc_append_rows.argtypes = [C.POINTER(Component), C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.orc_append_rows.restype = None
        L.orc_compute_image.argtypes = [C.POINTER(Component), C.c_int, C.POINTER(C.c_void_p), C.c_uint16, C.c_uint16,
                                        C.c_int, C.c_void_p, C.c_char_p]
        L.orc_compute_image.restype = C.c_int
        L.orc_ycbcr_to_rgb.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, C.c_void_p]
        L.orc_ycbcr_to_rgb.restype = None
        L.orc_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_uint16, C.c_uint16, C.c_int, C.c_int, C.POINTER(_Result)]
        L.orc_decode.restype = None
        L.orc_free_result.argtypes = [C.POINTER(_Result)]
        L.orc_free_result.restype = None
        L.orc_batch_pixels.argtypes = [C.POINTER(Component), C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_int,
                                       C.c_uint16, C.c_uint16, C.c_int, C.POINTER(C.c_void_p), C.c_int]
        L.orc_batch_pixels.restype = C.c_int
        L.orc_batch_decode.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong)]
        L.orc_batch_decode.restype = C.c_int
        _LIB = L
    return _LIB


class OracleError(Exception):
    def __init__(self, status, message):
        super().__init__(f"{STATUS.get(status, status)}: {message}")
        self.status = status
        self.kind = STATUS.get(status, str(status))


def idct_block(coefs, qt, scale=8, stride=None):
    """src/idct.rs:205-239 on one block -> (scale, scale) uint8 array."""
    coefs = np.ascontiguousarray(coefs, dtype=np.int16).reshape(64)
    qt = np.ascontiguousarray(qt, dtype=np.uint16).reshape(64)
    stride = stride or scale
    out = np.zeros(stride * scale, dtype=np.uint8)
    lib().orc_dequantize_and_idct_block(scale, coefs.ctypes.data, qt.ctypes.data, stride, out.ctypes.data)
    return out.reshape(scale, stride)[:, :scale].copy()


def make_components(width, height, sampling, dct_scale=8, tq=None, identifiers=None):
    """Frame geometry exactly as parse_sof + update_component_sizes compute it.
    sampling: list of (h, v). Returns (ctypes array of Component, (mcu_w, mcu_h))."""
    n = len(sampling)
    arr = (Component * n)()
    for i, (h, v) in enumerate(sampling):
        arr[i].identifier = identifiers[i] if identifiers else i + 1
        arr[i].h, arr[i].v = h, v
        arr[i].tq = tq[i] if tq else (0 if i == 0 else 1)
        arr[i].dct_scale = dct_scale
    mw, mh = C.c_uint16(), C.c_uint16()
    rc = lib().orc_update_component_sizes(width, height, arr, n, C.byref(mw), C.byref(mh))
    if rc:
        raise OracleError(rc, "invalid dimensions")
    return arr, (mw.value, mh.value)


def plane_bytes(comp):
    return int(lib().orc_plane_bytes(C.byref(comp)))


def idct_plane(comp, qt, coefs, n_mcu_rows=None):
    """ImmediateWorker: start + append_row x n + get_result for one component."""
    coefs = np.ascontiguousarray(coefs, dtype=np.int16).reshape(-1)
    qt = np.ascontiguousarray(qt, dtype=np.uint16).reshape(64)
    per_row = comp.block_w * comp.v * 64
    if n_mcu_rows is None:
        n_mcu_rows = coefs.size // per_row
    assert coefs.size >= n_mcu_rows * per_row
    plane = np.zeros(plane_bytes(comp), dtype=np.uint8)
    lib().orc_append_rows(C.byref(comp), qt.ctypes.data, coefs.ctypes.data, 0, n_mcu_rows, plane.ctypes.data)
    return plane


def compute_image(comps, planes, out_w, out_h, color_transform):
    """compute_image (src/decoder.rs:1300-1336) -> flat uint8 array."""
    n = len(planes)
    ct = CT[color_transform] if isinstance(color_transform, str) else int(color_transform)
    planes = [np.ascontiguousarray(p, dtype=np.uint8) for p in planes]
    ptrs = (C.c_void_p * n)(*[p.ctypes.data for p in planes])
    out_len = comps[0].size_w * comps[0].size_h if n == 1 else out_w * out_h * n
    out = np.zeros(out_len, dtype=np.uint8)
    msg = C.create_string_buffer(160)
    rc = lib().orc_compute_image(comps, n, ptrs, out_w, out_h, ct, out.ctypes.data, msg)
    if rc:
        raise OracleError(rc, msg.value.decode())
    return out


def pixels_from_coefficients(comps, qts, coefs, out_w, out_h, color_transform):
    planes = [idct_plane(comps[i], qts[i], coefs[i]) for i in range(len(coefs))]
    return compute_image(comps, planes, out_w, out_h, color_transform)


class Decoded:
    pass


def decode(data, scale_to=None, color_transform="AUTO", keep_intermediates=False):
    """Decoder::new(data) [+ scale(w,h)] [+ set_color_transform] + decode()."""
    data = bytes(data)
    res = _Result()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if data else (C.c_uint8 * 1)()
    rw, rh = scale_to if scale_to else (0, 0)
    ct = CT[color_transform] if isinstance(color_transform, str) else int(color_transform)
    lib().orc_decode(buf, len(data), rw, rh, ct, 1 if keep_intermediates else 0, C.byref(res))
    try:
        if res.status:
            raise OracleError(res.status, res.message.decode(errors="replace"))
        d = Decoded()
        d.width, d.height, d.ncomp = res.width, res.height, res.ncomp
        d.image_size = (res.image_w, res.image_h)
        d.coding_process = ("DctSequential", "DctProgressive", "Lossless")[res.coding_process]
        d.is_baseline = bool(res.is_baseline)
        d.color_transform = res.color_transform
        d.mcu_size = (res.mcu_w, res.mcu_h)
        d.components = (Component * res.ncomp)(*[res.components[i] for i in range(res.ncomp)])
        d.pixels = np.ctypeslib.as_array(res.pixels, shape=(res.pixels_len,)).copy() if res.pixels_len else np.zeros(0, np.uint8)
        d.pixel_format = {1: "L8", 3: "RGB24", 4: "CMYK32"}[res.ncomp]
        if keep_intermediates:
            d.coefs, d.planes, d.qtables = [], [], []
            for i in range(res.ncomp):
                n = res.coefs_len[i]
                d.coefs.append(np.ctypeslib.as_array(res.coefs[i], shape=(n,)).copy() if res.coefs[i] and n else None)
                m = res.planes_len[i]
                d.planes.append(np.ctypeslib.as_array(res.planes[i], shape=(m,)).copy() if res.have_plane[i] and m else None)
                d.qtables.append(np.array(list(res.qtables[i]), dtype=np.uint16))
        return d
    finally:
        lib().orc_free_result(C.byref(res))


def batch_pixels(comps, qts, coefs_per_image, out_w, out_h, color_transform, nthreads, keep_outputs=True):
    """CPU baseline leg: pixel pipeline for a batch of same-geometry images (oracle_batch.c).
    keep_outputs=False: timing mode, each thread decodes into one private reused buffer."""
    n = len(comps)
    n_images = len(coefs_per_image)
    qarr = np.ascontiguousarray(np.stack([np.asarray(q, dtype=np.uint16).reshape(64) for q in qts]))
    cptrs = (C.c_void_p * (n_images * n))()
    keep = []
    for i, per_comp in enumerate(coefs_per_image):
        for c in range(n):
            a = np.ascontiguousarray(per_comp[c], dtype=np.int16)
            keep.append(a)
            cptrs[i * n + c] = a.ctypes.data
    out_len = comps[0].size_w * comps[0].size_h if n == 1 else out_w * out_h * n
    outs = [np.zeros(out_len, dtype=np.uint8) for _ in range(n_images)] if keep_outputs else []
    optrs = (C.c_void_p * n_images)(*[o.ctypes.data for o in outs]) if keep_outputs else None
    ct = CT[color_transform] if isinstance(color_transform, str) else int(color_transform)
    rc = lib().orc_batch_pixels(comps, n, qarr.ctypes.data, cptrs, n_images, out_w, out_h, ct, optrs, nthreads)
    if rc:
        raise OracleError(rc, "batch")
    return outs


def batch_decode(streams, nthreads):
    """CPU baseline leg (end to end): Decoder::new(bytes).decode() of every stream, one stream per task over `nthreads`
    threads (oracle_batch.c).  -> (streams decoded, pixel bytes produced); the pixels themselves are dropped."""
    bufs = [bytes(s) for s in streams]
    n = len(bufs)
    ptrs = (C.c_char_p * max(n, 1))(*bufs)
    lens = (C.c_size_t * max(n, 1))(*[len(b) for b in bufs])
    ok, px = C.c_int(0), C.c_ulonglong(0)
    lib().orc_batch_decode(ptrs, lens, n, nthreads, C.byref(ok), C.byref(px))
    return ok.value, px.value
