"""ctypes handle on oracle/liboracle.so -- the CPU oracle (TEST INFRASTRUCTURE: checker / cpu_baseline only)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_float, c_int, c_int32, c_int64, c_uint8, c_uint16, c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(ORACLE_DIR, "avif_oracle.c")
    src2 = os.path.join(ORACLE_DIR, "cpu_baseline.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(src), os.path.getmtime(src2)):
        subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(LIB)
    f = c_float
    for name, n in [("oracle_linear_to_pq", 2), ("oracle_pq_to_linear", 2), ("oracle_linear_to_smpte428", 1),
                    ("oracle_smpte428_to_linear", 1), ("oracle_linear_to_hlg", 1), ("oracle_hlg_to_linear", 1)]:
        fn = getattr(L, name); fn.restype = f; fn.argtypes = [f] * n
    L.oracle_premultiply_f32.restype = f; L.oracle_premultiply_f32.argtypes = [f, f, f]
    L.oracle_unpremultiply_f32.restype = f; L.oracle_unpremultiply_f32.argtypes = [f, f, f]
    L.oracle_premultiply_u8.restype = c_uint8; L.oracle_premultiply_u8.argtypes = [c_uint8, c_uint8]
    L.oracle_unpremultiply_u8.restype = c_uint8; L.oracle_unpremultiply_u8.argtypes = [c_uint8, c_uint8]
    L.oracle_premultiply_u16.restype = c_uint16; L.oracle_premultiply_u16.argtypes = [c_uint16, c_uint16, c_uint16]
    L.oracle_unpremultiply_u16.restype = c_uint16; L.oracle_unpremultiply_u16.argtypes = [c_uint16, c_uint16, c_uint16]
    L.oracle_apply_hlg_ootf.restype = None; L.oracle_apply_hlg_ootf.argtypes = [POINTER(f * 3), POINTER(f * 3), f, f]
    L.oracle_apply_inverse_hlg_ootf.restype = None
    L.oracle_apply_inverse_hlg_ootf.argtypes = [POINTER(f * 3), POINTER(f * 3), f, f]
    L.oracle_hlg_luma_coefficients.restype = c_int; L.oracle_hlg_luma_coefficients.argtypes = [c_int32, POINTER(f * 3)]
    L.oracle_build_lut_8_to_n.restype = None; L.oracle_build_lut_8_to_n.argtypes = [c_int, c_void_p]
    L.oracle_build_lut_16_to_8.restype = None; L.oracle_build_lut_16_to_8.argtypes = [c_void_p]
    L.oracle_build_lut_16_to_n.restype = None; L.oracle_build_lut_16_to_n.argtypes = [c_int, c_void_p]
    L.oracle_limited_to_full_y.restype = c_int; L.oracle_limited_to_full_y.argtypes = [c_int, c_int]
    L.oracle_limited_to_full_uv.restype = c_int; L.oracle_limited_to_full_uv.argtypes = [c_int, c_int]
    L.oracle_build_yuv_tables.restype = c_int
    L.oracle_build_yuv_tables.argtypes = [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    L.oracle_get_yuv_coefficients.restype = None
    L.oracle_get_yuv_coefficients.argtypes = [c_int, c_int, c_int, POINTER(f * 3)]
    P4 = c_void_p * 4
    S4 = c_int64 * 4
    L.oracle_write_rows.restype = c_int32
    L.oracle_write_rows.argtypes = [c_void_p, c_int32, c_int32, c_void_p, c_int64, POINTER(P4), POINTER(S4)]
    L.oracle_read_rows.restype = c_int32
    L.oracle_read_rows.argtypes = [c_void_p, c_int32, c_int32, POINTER(P4), POINTER(S4), c_void_p, c_int64]
    L.oracle_write_image_row_callback.restype = c_int32
    L.oracle_write_image_row_callback.argtypes = [c_void_p, c_void_p, c_int64, POINTER(P4), POINTER(S4)]
    L.oracle_write_image_all_cores.restype = c_int32
    L.oracle_write_image_all_cores.argtypes = [c_void_p, c_void_p, c_int64, POINTER(P4), POINTER(S4), POINTER(c_int32)]
    L.oracle_read_image_all_cores.restype = c_int32
    L.oracle_read_image_all_cores.argtypes = [c_void_p, POINTER(P4), POINTER(S4), c_void_p, c_int64, POINTER(c_int32)]
    _lib = L
    return L
