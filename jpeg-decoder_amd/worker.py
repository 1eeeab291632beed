"""Host-side mirror of the reference's Worker boundary (src/worker/mod.rs:18-35,97-128) over
the C ABI — same names, argument meaning and error behaviour as the trait a Rust
``src/worker/hip.rs`` would implement."""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _native as N
from .error import check, FormatError

COLOR_TRANSFORMS = ("None", "Unknown", "Grayscale", "RGB", "YCbCr", "CMYK", "YCCK", "JcsBgYcc", "JcsBgRgb")


def color_transform_id(ct):
    """ColorTransform (src/decoder.rs:76-98) name or id -> id."""
    if isinstance(ct, str):
        low = {n.lower(): i for i, n in enumerate(COLOR_TRANSFORMS)}
        if ct.lower() not in low:
            raise FormatError(f"unknown colour transform {ct!r}")
        return low[ct.lower()]
    return int(ct)


# struct RowData, src/worker/mod.rs:18-22
RowData = namedtuple("RowData", "index component quantization_table")


class HipWorker:
    """``impl Worker`` backed by the MI355X kernels: planes stay in HBM between
    start / append_row / get_result and compute_image."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(N.lib().jpgpu_worker_create(device, C.byref(self._h)),
              b"no usable MI355X device (jpgpu_worker_create)")

    def close(self):
        if self._h:
            lib = N.lib() if N is not None and getattr(N, "lib", None) else None  # (interpreter shutdown: module globals may be gone)
            if lib is not None:
                lib.jpgpu_worker_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, status):
        check(status, N.lib().jpgpu_worker_last_error(self._h) if status else b"")

    def start(self, row_data):
        """Worker::start — src/worker/mod.rs:25"""
        qt = np.ascontiguousarray(row_data.quantization_table, dtype=np.uint16).reshape(64)
        self._check(N.lib().jpgpu_worker_start(self._h, row_data.index, C.byref(row_data.component), qt.ctypes.data))

    def append_row(self, row):
        """Worker::append_row((index, Vec<i16>)) — src/worker/mod.rs:26"""
        index, data = row
        data = np.ascontiguousarray(data, dtype=np.int16).reshape(-1)
        self._check(N.lib().jpgpu_worker_append_row(self._h, index, data.ctypes.data, data.size))

    def append_rows(self, rows):
        """Worker::append_rows(iterator of (index, Vec<i16>)) — src/worker/mod.rs:29-34"""
        for row in rows:
            self.append_row(row)

    def append_rows_contiguous(self, index, data, n_rows):
        data = np.ascontiguousarray(data, dtype=np.int16).reshape(-1)
        self._check(N.lib().jpgpu_worker_append_rows(self._h, index, data.ctypes.data, n_rows))

    def get_result(self, index, component=None):
        """Worker::get_result(index) -> Vec<u8> — src/worker/mod.rs:27"""
        n = C.c_size_t(0)
        # first call sizes the result
        cap = component.plane_bytes() if component is not None else 0
        if cap == 0:
            st = N.lib().jpgpu_worker_get_result(self._h, index, None, 0, C.byref(n))
            if n.value == 0:
                self._check(st)
                return np.zeros(0, dtype=np.uint8)
            cap = n.value
        out = np.empty(cap, dtype=np.uint8)
        self._check(N.lib().jpgpu_worker_get_result(self._h, index, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value]

    def finish_plane(self, index, plane_slot):
        """Device-resident get_result: keep the plane in HBM as frame component `plane_slot`."""
        self._check(N.lib().jpgpu_worker_finish_plane(self._h, index, plane_slot))

    @property
    def last_path(self):
        """Kernels of the last compute_image: "generic" or the fused kernel of the frame's kind."""
        return N.lib().jpgpu_worker_last_path(self._h).decode()

    @property
    def last_class(self):
        """jpgpu_worker_last_class: range class (0 / 1 / 3) the last fused compute_image ran with, -1 after the generic kernels."""
        return N.lib().jpgpu_worker_last_class(self._h)

    def compute_image(self, components, data, output_size, color_transform):
        """compute_image (src/decoder.rs:1300-1336) incl. compute_image_parallel
        (src/worker/mod.rs:97-128).  data: list of planes (np.uint8) or None to use the planes
        retained on the device."""
        ncomp = len(components)
        ptrs = None
        keep = []
        if data is not None:
            if len(data) == 0 or any(p is None or len(p) == 0 for p in data):
                raise FormatError("not all components have data")
            keep = [np.ascontiguousarray(p, dtype=np.uint8) for p in data]
            ptrs = (C.c_void_p * ncomp)(*[p.ctypes.data for p in keep])
        w, h = output_size
        out_len = (components[0].size_width * components[0].size_height) if ncomp == 1 else w * h * ncomp
        out = np.empty(max(out_len, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        comps = (N.Component * ncomp)(*components)
        self._check(N.lib().jpgpu_compute_image(self._h, comps, ncomp, ptrs, w, h, color_transform_id(color_transform),
                                                out.ctypes.data, out.size, C.byref(n)))
        return out[: n.value]


def compute_image_parallel(components, data, output_size, color_transform, device=0):
    """Free function of src/worker/mod.rs:97-128 (rayon twin src/worker/rayon.rs:193-219)."""
    with HipWorker(device) as w:
        return w.compute_image(components, data, output_size, color_transform)
