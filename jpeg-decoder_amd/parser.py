"""Frame geometry exactly as the reference computes it (src/parser.rs:12-16,76-89,282-310,
119-134).  Pure host logic, shared by the worker/batch mirrors, tests and bench."""
import math
from collections import namedtuple

from . import _native as N
from .error import FormatError

Dimensions = namedtuple("Dimensions", "width height")


def ceil_div(x, y):
    """src/parser.rs:282-290"""
    if x == 0 or y == 0:
        raise FormatError("invalid dimensions")
    return (1 + ((x - 1) // y)) & 0xFFFF


def update_component_sizes(size, components):
    """src/parser.rs:292-310 — fills size / block_size of every component, returns mcu_size."""
    h_max = max(c.horizontal_sampling_factor for c in components)
    v_max = max(c.vertical_sampling_factor for c in components)
    mcu = Dimensions(ceil_div(size.width, h_max * 8), ceil_div(size.height, v_max * 8))
    for c in components:
        c.size_width = ceil_div(size.width * c.horizontal_sampling_factor * c.dct_scale, h_max * 8)
        c.size_height = ceil_div(size.height * c.vertical_sampling_factor * c.dct_scale, v_max * 8)
        c.block_width = mcu.width * c.horizontal_sampling_factor
        c.block_height = mcu.height * c.vertical_sampling_factor
    return mcu


def make_components(width, height, sampling, dct_scale=8, identifiers=None, table_indices=None):
    """Components of a frame header: sampling = [(h, v), ...] (parse_sof, src/parser.rs:161-280)."""
    comps = (N.Component * len(sampling))()
    for i, (h, v) in enumerate(sampling):
        comps[i].identifier = identifiers[i] if identifiers else i + 1
        comps[i].horizontal_sampling_factor = h
        comps[i].vertical_sampling_factor = v
        comps[i].quantization_table_index = table_indices[i] if table_indices else (0 if i == 0 else 1)
        comps[i].dct_scale = dct_scale
    mcu = update_component_sizes(Dimensions(width, height), comps)
    return comps, mcu


def scaled_output_size(width, height, idct_size):
    """FrameInfo::update_idct_size, src/parser.rs:127-130 (f32 arithmetic)."""
    import numpy as np
    f = np.float32
    return (int(math.ceil(f(f(width) * f(idct_size)) / f(8.0))), int(math.ceil(f(f(height) * f(idct_size)) / f(8.0))))


def choose_idct_size(full, requested):
    """src/idct.rs:14-28"""
    def scaled(length, scale):
        return ((length * scale - 1) // 8 + 1) & 0xFFFF
    for scale in (1, 2, 4):
        if scaled(full.width, scale) >= requested.width or scaled(full.height, scale) >= requested.height:
            return scale
    return 8
