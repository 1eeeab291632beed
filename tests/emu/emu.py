"""TEST-ONLY: loader for the CPU emulation of the product's device code (tests/emu/*.cpp)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
        L = C.CDLL(os.path.join(_HERE, "libemu.so"))
        L.emu_idct8x8.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.emu_idct_small.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.emu_ycbcr.argtypes = [C.c_uint32] * 3
        L.emu_ycbcr.restype = C.c_uint32
        L.emu_ycbcr_centred_mismatches.argtypes = []
        L.emu_ycbcr_centred_mismatches.restype = C.c_uint32
        L.emu_fused_decode.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]
        L.emu_fused_decode.restype = C.c_int
        L.emu_compute_image.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.c_uint16, C.c_uint16, C.c_int, C.c_void_p,
                                        C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_int)]
        L.emu_compute_image.restype = C.c_int
        L.emu_scaled_fused.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.emu_scaled_fused.restype = C.c_int
        L.emu_huff_plan.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.emu_huff_covered.argtypes = [C.c_void_p, C.c_size_t]
        L.emu_huff_set_dri.argtypes = [C.c_uint32]
        L.emu_stage_segment.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.emu_stage_segment.restype = C.c_uint32
        L.emu_stage_segment_clean.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.emu_stage_segment_clean.restype = C.c_int
        L.emu_slot_bytes.argtypes = [C.c_uint32]
        L.emu_slot_bytes.restype = C.c_uint32
        L.emu_chunk_shift.argtypes = [C.c_uint32, C.c_uint32]
        L.emu_chunk_shift.restype = C.c_uint32
        L.emu_huff_set_launch.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.emu_huff_set_launch.restype = None
        L.emu_huff_decode.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        L.emu_prog_plan.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.emu_prog_decode.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.c_int]
        L.emu_prog_dependencies.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.emu_unstuff.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.emu_unstuff.restype = C.c_int
        _LIB = L
    return _LIB
