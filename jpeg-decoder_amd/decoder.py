"""Mirror of the crate's public ``Decoder`` (src/decoder.rs:101-295, re-exported by src/lib.rs:39-41):
``Decoder(reader).decode()``, ``read_info()``, ``info()``, ``scale()``, ``set_color_transform()``,
``set_max_decoding_buffer_size()``, ``exif_data()``, ``xmp_data()``, ``icc_profile()``.

Entropy decoding runs in the C++ host front-end, every pixel is produced on the MI355X."""
import ctypes as C
from collections import namedtuple

import numpy as np

from . import _native as N
from .error import check
from .worker import COLOR_TRANSFORMS, color_transform_id

PIXEL_FORMATS = ("L8", "L16", "RGB24", "CMYK32")  # src/decoder.rs:39-60
CODING_PROCESSES = ("DctSequential", "DctProgressive", "Lossless")  # src/parser.rs:24-33
PIXEL_BYTES = {"L8": 1, "L16": 2, "RGB24": 3, "CMYK32": 4}

# struct ImageInfo, src/decoder.rs:62-73
ImageInfo = namedtuple("ImageInfo", "width height pixel_format coding_process")


class Decoder:
    def __init__(self, reader, device=0):
        """Decoder::new(reader): `reader` is bytes-like or an object with .read()."""
        data = reader.read() if hasattr(reader, "read") else bytes(reader)
        self._buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\\0")
        self._h = C.c_void_p()
        check(N.lib().jpgpu_decoder_create(self._buf, len(data), device, C.byref(self._h)), b"jpgpu_decoder_create")

    def close(self):
        if getattr(self, "_h", None):
            lib = N.lib() if N is not None and getattr(N, "lib", None) else None  # (interpreter shutdown: module globals may be gone)
            if lib is not None:
                lib.jpgpu_decoder_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _check(self, st):
        check(st, N.lib().jpgpu_decoder_last_error(self._h) if st else b"")

    def set_color_transform(self, transform):
        self._check(N.lib().jpgpu_decoder_set_color_transform(self._h, color_transform_id(transform)))

    def set_max_decoding_buffer_size(self, max_bytes):
        self._check(N.lib().jpgpu_decoder_set_max_decoding_buffer_size(self._h, max_bytes))

    def read_info(self):
        self._check(N.lib().jpgpu_decoder_read_info(self._h))

    def info(self):
        """None until read_info() or decode() returned Ok (src/decoder.rs:170-197)."""
        i = N.ImageInfoStruct()
        if N.lib().jpgpu_decoder_info(self._h, C.byref(i)):
            return None
        return ImageInfo(i.width, i.height, PIXEL_FORMATS[i.pixel_format], CODING_PROCESSES[i.coding_process])

    def scale(self, requested_width, requested_height):
        w, h = C.c_uint16(), C.c_uint16()
        self._check(N.lib().jpgpu_decoder_scale(self._h, requested_width, requested_height, C.byref(w), C.byref(h)))
        return w.value, h.value

    def decode(self):
        """decode() -> Vec<u8> (a numpy uint8 array)."""
        n = C.c_size_t(0)
        cap = N.lib().jpgpu_decoder_output_bytes(self._h)
        if cap == 0:  # size the buffer from the frame header first: otherwise a large image is decoded into the library's
            N.lib().jpgpu_decoder_read_info(self._h)  # own buffer and copied twice (an error here is decode()'s too: it raises it)
            cap = N.lib().jpgpu_decoder_output_bytes(self._h)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        st = N.lib().jpgpu_decoder_decode(self._h, out.ctypes.data, cap, C.byref(n))
        if st and n.value > cap:  # size only known after the frame header was parsed
            cap = n.value
            out = np.empty(max(cap, 1), dtype=np.uint8)
            st = N.lib().jpgpu_decoder_decode(self._h, out.ctypes.data, cap, C.byref(n))
        self._check(st)
        return out[: n.value]

    def _blob(self, fn):
        n = C.c_size_t(0)
        p = fn(self._h, C.byref(n))
        return None if not p else C.string_at(p, n.value)

    def exif_data(self):
        return self._blob(N.lib().jpgpu_decoder_exif_data)

    def xmp_data(self):
        return self._blob(N.lib().jpgpu_decoder_xmp_data)

    def icc_profile(self):
        return self._blob(N.lib().jpgpu_decoder_icc_profile)

    def decode_coefficients(self):
        """Host half only: (ImageDesc, [np.int16 coefficient rows per component]) — what crosses
        the Worker boundary; feeds Batch.upload when many files are decoded together."""
        desc = N.ImageDesc()
        ptrs = (C.c_void_p * 4)()
        ns = (C.c_size_t * 4)()
        self._check(N.lib().jpgpu_decoder_decode_coefficients(self._h, C.byref(desc), ptrs, ns))
        coefs = []
        for c in range(desc.ncomp):
            n = ns[c]
            coefs.append(np.ctypeslib.as_array(C.cast(ptrs[c], C.POINTER(C.c_int16)), shape=(n,)).copy() if n else np.zeros(0, np.int16))
        return desc, coefs


def decode_batch(files, device=0):
    """Decode many JPEG byte strings: host entropy decoding per file, then ONE batch of kernels.
    Returns a list of (ImageInfo, pixels)."""
    from .batch import Batch

    decs = [Decoder(f, device=-1) for f in files]
    descs, all_coefs, infos = [], [], []
    for d in decs:
        desc, coefs = d.decode_coefficients()
        descs.append(desc)
        all_coefs.append(coefs)
        infos.append(d.info())
    b = Batch(descs, device=device)
    for i, coefs in enumerate(all_coefs):
        for c, a in enumerate(coefs):
            full = b.coef_bytes(i, c) // 2
            if a.size < full:  # rows the scan never reached stay zero coefficients
                a = np.concatenate([a, np.zeros(full - a.size, np.int16)])
            b.upload(i, c, a)
    b.decode()
    b.synchronize()
    out = [(infos[i], b.download(i)) for i in range(len(files))]
    b.close()
    for d in decs:
        d.close()
    return out
