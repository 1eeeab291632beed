"""Batch driver mirror (include/jpgpu.h jpgpu_batch_*): N independent images per launch."""
import ctypes as C

import numpy as np

from . import _native as N
from .error import check
from .worker import color_transform_id


def image_desc(components, quantization_tables, out_w, out_h, color_transform):
    d = N.ImageDesc()
    d.ncomp = len(components)
    for i, c in enumerate(components):
        d.components[i] = c
        q = np.ascontiguousarray(quantization_tables[i], dtype=np.uint16).reshape(64)
        for k in range(64):
            d.quantization_tables[i][k] = int(q[k])
    d.out_w, d.out_h = out_w, out_h
    d.color_transform = color_transform_id(color_transform)
    return d


class Batch:
    def __init__(self, descs, device=0, flags=N.BATCH_DEFAULT):
        self._h = C.c_void_p()
        arr = (N.ImageDesc * len(descs))(*descs)
        st = N.lib().jpgpu_batch_create(device, arr, len(descs), flags, C.byref(self._h))
        if st:
            msg = N.lib().jpgpu_batch_last_error(self._h) if self._h else b"jpgpu_batch_create"
            msg = bytes(msg)
            self.close()
            check(st, msg)
        self.n_images = len(descs)
        self.descs = descs

    def close(self):
        if getattr(self, "_h", None):
            lib = N.lib() if N is not None and getattr(N, "lib", None) else None  # (interpreter shutdown: module globals may be gone)
            if lib is not None:
                lib.jpgpu_batch_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _check(self, st):
        check(st, N.lib().jpgpu_batch_last_error(self._h) if st else b"")

    @property
    def path(self):
        return N.lib().jpgpu_batch_path(self._h).decode()

    def coef_arena_bytes(self):
        return N.lib().jpgpu_batch_coef_arena_bytes(self._h)

    def out_arena_bytes(self):
        return N.lib().jpgpu_batch_out_arena_bytes(self._h)

    def coef_offset(self, image, comp):
        return N.lib().jpgpu_batch_coef_offset(self._h, image, comp)

    def coef_bytes(self, image, comp):
        return N.lib().jpgpu_batch_coef_bytes(self._h, image, comp)

    def out_offset(self, image):
        return N.lib().jpgpu_batch_out_offset(self._h, image)

    def out_bytes(self, image):
        return N.lib().jpgpu_batch_out_bytes(self._h, image)

    def bind(self, coef_ptr, out_ptr):
        self._check(N.lib().jpgpu_batch_bind(self._h, coef_ptr, out_ptr))

    def coef_arena(self):
        return N.lib().jpgpu_batch_coef_arena(self._h)

    def out_arena(self):
        return N.lib().jpgpu_batch_out_arena(self._h)

    def set_quantization_table(self, image, comp, table):
        """jpgpu_batch_set_quantization_table: replace the descriptor's table (the component's range class becomes 0 if it changes)."""
        q = np.ascontiguousarray(table, dtype=np.uint16).reshape(64)
        self._check(N.lib().jpgpu_batch_set_quantization_table(self._h, image, comp, q.ctypes.data))
        for k in range(64):
            self.descs[image].quantization_tables[comp][k] = int(q[k])

    def upload(self, image, comp, coefficients):
        a = np.ascontiguousarray(coefficients, dtype=np.int16).reshape(-1)
        self._check(N.lib().jpgpu_batch_upload(self._h, image, comp, a.ctypes.data, a.size))

    def upload_compact(self, image, comp, coefficients, stream=None, classify=True):
        """Same result as upload(), but PCIe carries only the non-zero coefficients (bitmap + index + values per block,
        include/jpgpu.h); a kernel expands them into the arena at the start of the next decode().  classify=False sends
        range_class = -1: the expansion kernel ranges the values on the device instead of the host encoder."""
        a = np.ascontiguousarray(coefficients, dtype=np.int16).reshape(-1)
        q = np.ascontiguousarray(np.ctypeslib.as_array(self.descs[image].quantization_tables[comp]), dtype=np.uint16)
        buf = np.empty(N.lib().jpgpu_compact_max_bytes(a.size // 64), np.uint8)
        rc = C.c_int(0)
        n = N.lib().jpgpu_compact_encode(a.ctypes.data, a.size // 64, q.ctypes.data, buf.ctypes.data, C.byref(rc))
        self._check(N.lib().jpgpu_batch_upload_compact(self._h, image, comp, buf.ctypes.data, n, rc.value if classify else -1, stream))
        self.synchronize(stream)  # the host buffer is pageable and about to go away
        return n

    def set_range_hint(self, image, range_class):
        """0 unknown/hostile, 1 every |c*q| < 2^15, 3 additionally every block-column sum of |c*q| <= 5900."""
        self._check(N.lib().jpgpu_batch_set_range_hint(self._h, image, int(range_class)))

    def scan_ranges(self, stream=None):
        """Range classes from the coefficients as they stand in the device arena (one pass at HBM speed): [image][comp]."""
        out = np.zeros((self.n_images, 4), np.uint8)
        self._check(N.lib().jpgpu_batch_scan_ranges(self._h, stream, out.ctypes.data))
        return out

    def classify_on_device(self, stream=None):
        """jpgpu_batch_classify_on_device: range statistics of the arena's coefficients gathered and kept ON THE DEVICE
        (asynchronous, no read-back); decode() then takes the classes from them there.  The device entropy decoder and
        upload_compact(..., classify=False) leave the same statistics as a by-product."""
        self._check(N.lib().jpgpu_batch_classify_on_device(self._h, stream))

    def class_counts(self):
        """Images of the fused launch groups per arithmetic variant: (wrap-exact, range class 1, range class 3)."""
        c = (C.c_uint32 * 3)()
        self._check(N.lib().jpgpu_batch_class_counts(self._h, c))
        return tuple(int(x) for x in c)

    def decode(self, stream=None):
        self._check(N.lib().jpgpu_batch_decode(self._h, stream))

    def synchronize(self, stream=None):
        self._check(N.lib().jpgpu_batch_synchronize(self._h, stream))

    def download(self, image):
        n = self.out_bytes(image)
        out = np.empty(max(n, 1), dtype=np.uint8)
        got = C.c_size_t(0)
        self._check(N.lib().jpgpu_batch_download(self._h, image, out.ctypes.data, out.size, C.byref(got)))
        return out[: got.value]

    def time(self, iters, stream=None):
        ms = C.c_float(0)
        self._check(N.lib().jpgpu_batch_time(self._h, stream, iters, C.byref(ms)))
        return ms.value
