"""ctypes binding of libjpgpu.so (the C ABI declared in include/jpgpu.h).

There is no CPU fallback anywhere in this package: if the HIP library is missing or no
MI355X is visible, compute entry points raise.  PyTorch is not needed by the product path
(only bench.py uses it for torch.distributed and device tensors)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)


def process_init():
    """Opt-in (jpgpu_process_init of include/jpgpu.h, done here without loading the library): GPU_MAX_HW_QUEUES=24 unless set.
    jpgpu_pipeline_decode keeps several sub-batches in flight on 12 compute + 4 copy + 2 download streams; the HIP runtime maps a process's
    streams onto GPU_MAX_HW_QUEUES hardware queues (default 4: streams that share one serialise — 4,096 1080p files per call
    91 ms with 4 queues, 64 ms with 16).  The runtime reads the variable once, when it initialises: call this before the first
    HIP call of the process.  Importing the package no longer does it (bench.py, tools/ and tests/conftest.py call it)."""
    if "GPU_MAX_HW_QUEUES" in os.environ:  # (like the C function: True only if THIS call set the variable)
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = "24"
    return True


# JPGPU_LIBRARY: development knob for A/B builds of the same ABI (e.g. libjpgpu_alt.so built with other -D flags)
LIB_PATH = os.environ.get("JPGPU_LIBRARY") or os.path.join(_HERE, "libjpgpu.so")
HEADER_PATH = os.path.join(_ROOT, "include", "jpgpu.h")

OK, ERR_FORMAT, ERR_UNSUPPORTED, ERR_IO, ERR_INTERNAL, ERR_NO_DEVICE = range(6)
MAX_COMPONENTS = 4

BATCH_DEFAULT, BATCH_EXTERNAL_BUFFERS, BATCH_FORCE_GENERIC, BATCH_ASSUME_HOSTILE = 0, 1, 2, 4


class Component(C.Structure):
    """parser::Component (src/parser.rs:76-89) as laid out in include/jpgpu.h."""
    _fields_ = [
        ("identifier", C.c_uint8),
        ("horizontal_sampling_factor", C.c_uint8),
        ("vertical_sampling_factor", C.c_uint8),
        ("quantization_table_index", C.c_uint8),
        ("dct_scale", C.c_uint32),
        ("size_width", C.c_uint16),
        ("size_height", C.c_uint16),
        ("block_width", C.c_uint16),
        ("block_height", C.c_uint16),
    ]

    def plane_bytes(self):
        return self.block_width * self.block_height * self.dct_scale * self.dct_scale

    def coefficient_count(self):
        return self.block_width * self.block_height * 64


class ImageDesc(C.Structure):
    _fields_ = [
        ("ncomp", C.c_uint32),
        ("components", Component * 4),
        ("quantization_tables", (C.c_uint16 * 64) * 4),
        ("out_w", C.c_uint16),
        ("out_h", C.c_uint16),
        ("color_transform", C.c_int32),
    ]


class PipelineTimings(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("headers_ms", "setup_ms", "entropy_and_upload_ms", "kernels_ms", "download_ms", "total_ms")] + \
               [("threads", C.c_uint32), ("images_ok", C.c_uint32), ("jpeg_bytes", C.c_uint64), ("coefficient_bytes", C.c_uint64),
                ("pixel_bytes", C.c_uint64), ("images_device_entropy", C.c_uint32), ("images_device_rejected", C.c_uint32),
                ("dev_times_valid", C.c_uint32), ("_pad", C.c_uint32)] + \
               [(n, C.c_double) for n in ("dev_fill_ms", "dev_sync_ms", "dev_write_ms", "dev_pixel_ms", "decode_ms", "gather_ms")] + \
               [("gather_bytes", C.c_uint64), ("gather_copy_ms", C.c_double), ("cpu_ms", C.c_double), ("images_host_light", C.c_uint32),
                ("input_pinned", C.c_uint32), ("images_device_progressive", C.c_uint32), ("images_entry_pixels", C.c_uint32)]


PIPELINE_DOWNLOAD, PIPELINE_DENSE, PIPELINE_DEVICE_ENTROPY, PIPELINE_GATHER = 1, 2, 4, 16
PIPELINE_HOST_LIGHT, PIPELINE_HOST_STAGED, PIPELINE_INPUT_PINNED, PIPELINE_PROGRESSIVE_ON_HOST = 32, 64, 128, 256
PIPELINE_MULTI_PIN_CPUS = 1


class ImageInfoStruct(C.Structure):
    _fields_ = [("width", C.c_uint16), ("height", C.c_uint16), ("pixel_format", C.c_int32), ("coding_process", C.c_int32)]


def build(force=False, verbose=False):
    """Compile libjpgpu.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(dp, f) for dp, _, fs in os.walk(csrc) for f in fs if f.endswith((".cpp", ".hip", ".hpp"))]
    srcs += [HEADER_PATH, os.path.join(_ROOT, "include", "jpgpu_decoder.h")]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        cmd = ["make", "-C", csrc, "-j8"] + ([] if verbose else ["-s"])
        subprocess.check_call(cmd)
    return LIB_PATH


_LIB = None

_PROTOS = {
    # name: (restype, argtypes)
    "jpgpu_version": (C.c_char_p, []),
    "jpgpu_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "jpgpu_status_string": (C.c_char_p, [C.c_int]),
    "jpgpu_process_init": (C.c_int, []),
    "jpgpu_worker_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "jpgpu_worker_destroy": (None, [C.c_void_p]),
    "jpgpu_worker_last_error": (C.c_char_p, [C.c_void_p]),
    "jpgpu_worker_last_path": (C.c_char_p, [C.c_void_p]),
    "jpgpu_worker_start": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(Component), C.c_void_p]),
    "jpgpu_worker_append_row": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]),
    "jpgpu_worker_append_rows": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]),
    "jpgpu_worker_get_result": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "jpgpu_worker_finish_plane": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "jpgpu_worker_last_class": (C.c_int, [C.c_void_p]),
    "jpgpu_compute_image": (C.c_int, [C.c_void_p, C.POINTER(Component), C.c_uint32, C.POINTER(C.c_void_p), C.c_uint16,
                                      C.c_uint16, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "jpgpu_batch_create": (C.c_int, [C.c_int, C.POINTER(ImageDesc), C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "jpgpu_batch_destroy": (None, [C.c_void_p]),
    "jpgpu_batch_last_error": (C.c_char_p, [C.c_void_p]),
    "jpgpu_batch_coef_arena_bytes": (C.c_size_t, [C.c_void_p]),
    "jpgpu_batch_out_arena_bytes": (C.c_size_t, [C.c_void_p]),
    "jpgpu_batch_coef_offset": (C.c_size_t, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "jpgpu_batch_coef_bytes": (C.c_size_t, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "jpgpu_batch_out_offset": (C.c_size_t, [C.c_void_p, C.c_uint32]),
    "jpgpu_batch_out_bytes": (C.c_size_t, [C.c_void_p, C.c_uint32]),
    "jpgpu_batch_bind": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "jpgpu_batch_coef_arena": (C.c_void_p, [C.c_void_p]),
    "jpgpu_batch_out_arena": (C.c_void_p, [C.c_void_p]),
    "jpgpu_batch_upload": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]),
    "jpgpu_batch_set_range_hint": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int]),
    "jpgpu_batch_set_range_class": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]),
    "jpgpu_batch_scan_ranges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "jpgpu_batch_classify_on_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jpgpu_range_class": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "jpgpu_batch_set_quantization_table": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "jpgpu_compact_max_bytes": (C.c_size_t, [C.c_size_t]),
    "jpgpu_compact_encode": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "jpgpu_batch_upload_compact": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "jpgpu_batch_decode": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jpgpu_batch_synchronize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jpgpu_batch_download": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "jpgpu_batch_time": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]),
    "jpgpu_batch_path": (C.c_char_p, [C.c_void_p]),
    "jpgpu_batch_class_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    # include/jpgpu_decoder.h
    "jpgpu_decoder_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "jpgpu_decoder_destroy": (None, [C.c_void_p]),
    "jpgpu_decoder_last_error": (C.c_char_p, [C.c_void_p]),
    "jpgpu_decoder_set_color_transform": (C.c_int, [C.c_void_p, C.c_int]),
    "jpgpu_decoder_set_max_decoding_buffer_size": (C.c_int, [C.c_void_p, C.c_size_t]),
    "jpgpu_decoder_read_info": (C.c_int, [C.c_void_p]),
    "jpgpu_decoder_info": (C.c_int, [C.c_void_p, C.POINTER(ImageInfoStruct)]),
    "jpgpu_decoder_scale": (C.c_int, [C.c_void_p, C.c_uint16, C.c_uint16, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]),
    "jpgpu_decoder_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "jpgpu_decoder_output_bytes": (C.c_size_t, [C.c_void_p]),
    "jpgpu_decoder_exif_data": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "jpgpu_decoder_xmp_data": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "jpgpu_decoder_icc_profile": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "jpgpu_decoder_decode_coefficients": (C.c_int, [C.c_void_p, C.POINTER(ImageDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "jpgpu_pipeline_create": (C.c_int, [C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]),
    "jpgpu_pipeline_create_multi": (C.c_int, [C.POINTER(C.c_int), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "jpgpu_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "jpgpu_host_free": (None, [C.c_void_p]),
    "jpgpu_plan_cpu_shares": (C.c_int, [C.c_char_p, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "jpgpu_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "jpgpu_pipeline_device_count": (C.c_uint32, [C.c_void_p]),
    "jpgpu_pipeline_image_device": (C.c_int, [C.c_void_p, C.c_uint32]),
    "jpgpu_pipeline_pixels_device_ordinal": (C.c_int, [C.c_void_p, C.c_uint32]),
    "jpgpu_pipeline_destroy": (None, [C.c_void_p]),
    "jpgpu_pipeline_last_error": (C.c_char_p, [C.c_void_p]),
    "jpgpu_trim_caches": (None, []),
    "jpgpu_pipeline_decode": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_uint32, C.c_uint32]),
    "jpgpu_pipeline_image_status": (C.c_int, [C.c_void_p, C.c_uint32]),
    "jpgpu_pipeline_image_error": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "jpgpu_pipeline_image_info": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(ImageInfoStruct)]),
    "jpgpu_pipeline_pixel_bytes": (C.c_size_t, [C.c_void_p, C.c_uint32]),
    "jpgpu_pipeline_pixels_device": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "jpgpu_pipeline_pixels_host": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "jpgpu_pipeline_kernel_path": (C.c_char_p, [C.c_void_p]),
    "jpgpu_pipeline_set_scale": (C.c_int, [C.c_void_p, C.c_uint16, C.c_uint16]),
    "jpgpu_pipeline_set_color_transform": (C.c_int, [C.c_void_p, C.c_int]),
    "jpgpu_pipeline_set_max_decoding_buffer_size": (C.c_int, [C.c_void_p, C.c_size_t]),
    "jpgpu_pipeline_download": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "jpgpu_pipeline_last_timings": (C.c_int, [C.c_void_p, C.POINTER(PipelineTimings)]),
}


def exported_symbols():
    return sorted(_PROTOS)


def lib():
    """Load libjpgpu.so (must have been built: `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950). "
                               "There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def device_count():
    n = C.c_int(0)
    lib().jpgpu_device_count(C.byref(n))
    return n.value
