"""Many JPEG streams -> pixels: ``Pipeline(device, threads).decode([bytes, ...])``.

N reference ``Decoder``s (src/decoder.rs:134-154, 293-295) run as one pipeline: headers and entropy decoding on a host
thread pool (one image per task), per-image asynchronous upload, one batch of kernels on the MI355X
(include/jpgpu_decoder.h, jpgpu_pipeline_*).  A failed image yields its ``Error`` in the result list instead of
pixels, the others are unaffected — as with independent decoders."""
import ctypes as C

import numpy as np

from . import _native as N
from .decoder import CODING_PROCESSES, PIXEL_FORMATS, ImageInfo
from .error import check, error_for_status
from .worker import color_transform_id


class PinnedFiles:
    """JPEG files in page-locked host memory (jpgpu_host_alloc), one after the other with `gap` bytes between them: what a loader
    that reads its files straight into a pinned arena holds.  ``Pipeline.decode(PinnedFiles(...), input_pinned=True)`` lets the DMA
    engine read the arena itself — no staging copy on the host."""

    def __init__(self, files, gap=64):
        files = [bytes(f) for f in files]
        self.lengths = [len(f) for f in files]
        self.offsets, total = [], 0
        for n in self.lengths:
            self.offsets.append(total)
            total += -(-(n + gap) // 64) * 64
        self._p = C.c_void_p()
        check(N.lib().jpgpu_host_alloc(max(total, 1), C.byref(self._p)), b"jpgpu_host_alloc")
        self.nbytes = total
        arena = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_uint8)), shape=(max(total, 1),))
        arena[:] = 0
        for f, o in zip(files, self.offsets):
            arena[o:o + len(f)] = np.frombuffer(f, np.uint8)
        self.base = self._p.value

    def __len__(self):
        return len(self.lengths)

    def close(self):
        if getattr(self, "_p", None):
            lib = N.lib() if N is not None and getattr(N, "lib", None) else None
            if lib is not None:
                lib.jpgpu_host_free(self._p)
            self._p = C.c_void_p()

    __del__ = close


class Pipeline:
    def __init__(self, device=0, threads=0, devices=None, pin_cpus=False):
        """device: one HIP ordinal; devices=[...]: jpgpu_pipeline_create_multi — image i of a call goes to devices[i mod n] (an ordinal
        may appear more than once), `threads` is then the host-thread budget of ALL devices together; pin_cpus: every device's
        threads on their own share of the CPUs this process may use."""
        self._h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            st = N.lib().jpgpu_pipeline_create_multi(arr, len(devices), threads, (N.PIPELINE_MULTI_PIN_CPUS if pin_cpus is True else int(pin_cpus or 0)), C.byref(self._h))
        else:
            st = N.lib().jpgpu_pipeline_create(device, threads, C.byref(self._h))
        if st:
            msg = N.lib().jpgpu_pipeline_last_error(self._h) if self._h else b"jpgpu_pipeline_create"
            self.close()
            check(st, msg)

    def close(self):
        if getattr(self, "_h", None):
            lib = N.lib() if N is not None and getattr(N, "lib", None) else None  # (interpreter shutdown: module globals may be gone)
            if lib is not None:
                lib.jpgpu_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def decode(self, streams, download=True, dense=False, device_entropy=True, scale=None, color_transform=None, max_decoding_buffer_size=None,
               gather=False, host_light=None, input_pinned=False, progressive_on_host=False):
        """-> list with, per stream, a numpy uint8 array of the decoded pixels (``Decoder.decode()``'s Vec<u8>) or the
        ``Error`` instance that stream produced.  download=False leaves the pixels in HBM (see ``device_pointer``); dense=True sends all
        64 coefficients of every block over PCIe instead of the compact form (same pixels, A/B switch); device_entropy=True
        (the default here; JPGPU_PIPELINE_DEVICE_ENTROPY in the C API) decodes 8-bit sequential Huffman streams (one scan with all components; with or without restart markers) on the
        GPU, all other streams — and any the device decoder flags — on the host as usual; scale=(w, h): every image as
        after ``Decoder.scale(w, h)`` (the smallest DCT scale whose output is at least w x h; ``info(i)`` gives the scaled size);
        color_transform: every image as after ``Decoder.set_color_transform(...)`` ("None", "Grayscale", "RGB", "YCbCr", "CMYK", "YCCK");
        max_decoding_buffer_size: ``Decoder.set_max_decoding_buffer_size`` (images that would need more fail with the reference's error);
        gather=True (pipelines over several devices): copy every device's pixels to the first device, per sub-batch behind its kernels (JPGPU_PIPELINE_GATHER);
        download="pinned": copy the pixels to the pipeline's pinned host buffers but return byte counts — look at them with ``pixels_host(i)``
        (no Python copy per image); host_light=True / False: JPGPU_PIPELINE_HOST_LIGHT / _HOST_STAGED (None: the library's default — host light,
        at every thread count); streams may be a ``PinnedFiles`` arena, with input_pinned=True the device reads it directly;
        progressive_on_host=True: progressive frames on the host entropy decoder even with device_entropy (A/B)."""
        L = N.lib()
        check(L.jpgpu_pipeline_set_max_decoding_buffer_size(self._h, (1 << 64) - 1 if max_decoding_buffer_size is None else int(max_decoding_buffer_size)), b"set_max")
        check(L.jpgpu_pipeline_set_color_transform(self._h, color_transform_id(color_transform) if color_transform is not None else -1), b"set_color_transform")
        check(L.jpgpu_pipeline_set_scale(self._h, *((int(scale[0]), int(scale[1])) if scale else (0, 0))), b"set_scale")
        if isinstance(streams, PinnedFiles):
            n = len(streams)
            ptrs = (C.c_void_p * max(n, 1))(*[streams.base + o for o in streams.offsets])
            lens = (C.c_size_t * max(n, 1))(*streams.lengths)
        else:
            if input_pinned:
                raise ValueError("input_pinned=True needs a PinnedFiles arena")
            bufs = [bytes(s.read() if hasattr(s, "read") else s) for s in streams]
            n = len(bufs)
            ptrs = (C.c_char_p * max(n, 1))(*bufs)  # the bytes objects' own buffers (alive in `bufs` during the call): no copies
            lens = (C.c_size_t * max(n, 1))(*[len(b) for b in bufs])
        keep_pinned = download == "pinned"
        flags = ((N.PIPELINE_DOWNLOAD if download else 0) | (N.PIPELINE_DENSE if dense else 0) | (N.PIPELINE_DEVICE_ENTROPY if device_entropy else 0) |
                 (N.PIPELINE_GATHER if gather else 0) | (N.PIPELINE_INPUT_PINNED if input_pinned else 0) | (N.PIPELINE_PROGRESSIVE_ON_HOST if progressive_on_host else 0) |
                 (0 if host_light is None else (N.PIPELINE_HOST_LIGHT if host_light else N.PIPELINE_HOST_STAGED)))
        st = L.jpgpu_pipeline_decode(self._h, C.cast(ptrs, C.POINTER(C.c_void_p)), lens, n, flags)
        check(st, L.jpgpu_pipeline_last_error(self._h) if st else b"")
        out = []
        for i in range(n):
            s = L.jpgpu_pipeline_image_status(self._h, i)
            if s:
                out.append(error_for_status(s, L.jpgpu_pipeline_image_error(self._h, i)))
                continue
            nbytes = L.jpgpu_pipeline_pixel_bytes(self._h, i)
            if download and not keep_pinned:
                p = L.jpgpu_pipeline_pixels_host(self._h, i)
                out.append(np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,)).copy() if nbytes else
                           np.zeros(0, np.uint8))
            else:
                out.append(nbytes)
        return out

    def info(self, image):
        i = N.ImageInfoStruct()
        if N.lib().jpgpu_pipeline_image_info(self._h, image, C.byref(i)):
            return None
        return ImageInfo(i.width, i.height, PIXEL_FORMATS[i.pixel_format], CODING_PROCESSES[i.coding_process])

    def download(self, image):
        """jpgpu_pipeline_download: one image's pixels of the last call from HBM (for calls made with download=False)."""
        n = N.lib().jpgpu_pipeline_pixel_bytes(self._h, image)
        out = np.empty(max(n, 1), np.uint8)
        got = C.c_size_t(0)
        st = N.lib().jpgpu_pipeline_download(self._h, image, out.ctypes.data, out.size, C.byref(got))
        check(st, N.lib().jpgpu_pipeline_last_error(self._h) if st else b"")
        return out[: got.value]

    def pixels_host(self, image):
        """View (no copy) of one image's pixels in the pipeline's pinned host buffer after a call with download=True / "pinned";
        valid until the next call."""
        n = N.lib().jpgpu_pipeline_pixel_bytes(self._h, image)
        p = N.lib().jpgpu_pipeline_pixels_host(self._h, image)
        if not p or not n:
            return None
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,))

    def device_pointer(self, image):
        return N.lib().jpgpu_pipeline_pixels_device(self._h, image)

    def device_of(self, image):
        """(device that decoded the image, device whose memory device_pointer(image) points into) — the same unless gathered."""
        return N.lib().jpgpu_pipeline_image_device(self._h, image), N.lib().jpgpu_pipeline_pixels_device_ordinal(self._h, image)

    @property
    def n_devices(self):
        return N.lib().jpgpu_pipeline_device_count(self._h)

    @property
    def kernel_path(self):
        return N.lib().jpgpu_pipeline_kernel_path(self._h).decode()

    def timings(self):
        t = N.PipelineTimings()
        check(N.lib().jpgpu_pipeline_last_timings(self._h, C.byref(t)), b"timings")
        return {name: getattr(t, name) for name, _ in N.PipelineTimings._fields_}
