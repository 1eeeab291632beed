"""jpeg-decoder_amd — MI355X (gfx950) pixel-pipeline backend for image-rs/jpeg-decoder.

Python face of the C ABI in include/jpgpu.h: the Worker boundary (HipWorker,
compute_image_parallel), the batch driver and the Decoder front-end.  The product path is
HIP-only: nothing here falls back to a CPU implementation."""
from . import _native
from ._native import Component, ImageDesc, build, device_count, lib, process_init
from .batch import Batch, image_desc
from .decoder import CODING_PROCESSES, PIXEL_FORMATS, Decoder, ImageInfo, decode_batch
from .error import Error, FormatError, InternalError, IoError, NoDeviceError, UnsupportedError
from .pipeline import PinnedFiles, Pipeline
from .parser import Dimensions, choose_idct_size, make_components, scaled_output_size, update_component_sizes
from .worker import COLOR_TRANSFORMS, HipWorker, RowData, color_transform_id, compute_image_parallel

__all__ = [
    "Batch", "CODING_PROCESSES", "COLOR_TRANSFORMS", "Component", "Decoder", "Dimensions", "ImageInfo", "PIXEL_FORMATS", "PinnedFiles", "Pipeline", "decode_batch", "Error", "FormatError", "HipWorker", "ImageDesc",
    "InternalError", "IoError", "NoDeviceError", "RowData", "UnsupportedError", "build", "choose_idct_size",
    "color_transform_id", "compute_image_parallel", "device_count", "image_desc", "lib", "make_components",
    "scaled_output_size", "update_component_sizes",
]
