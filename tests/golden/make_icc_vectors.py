#!/usr/bin/env python3
"""Generates tests/golden/icc_vectors.npz with the REAL Little CMS 2 of this image (lcms2 2.12 under /opt/conda, driven like the
reference by oracle/icc_oracle.c): for a handful of document profiles, a few thousand input pixels per document depth and what
lcms2 returns for them.  The fixture lets the ICC parity tests run where lcms2 is absent; where it is present the same tests
also run against the live library (tests/test_icc8.py, test_icc16.py, test_gpu_icc.py)."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "..", "..", "oracle", "liboracle_icc.so"))
L.oracle_icc_make_profile.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_void_p, ctypes.c_uint32]
A = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
L.oracle_icc_convert_rows_to_srgb8.argtypes = A
L.oracle_icc_convert_rows_to_rec2020.argtypes = A
L.oracle_icc_convert_rows_to_srgb_float.argtypes = A
L.oracle_icc_convert_rows_to_srgb16.argtypes = A[:3] + [ctypes.c_int32] + A[3:]

PROFILES = {"adobergb_g22": (3, 0, 2.19921875), "p3_srgbtrc": (1, 1, 0.0), "prophoto_d50_g18": (2, 0, 1.8),
            "p3_linear": (1, 0, 1.0), "p3_sampled1024": (1, 3, 1024.0)}
N = 1024


def main():
    out = {}
    rng = np.random.default_rng(2024)
    for name, (kind, trc, g) in PROFILES.items():
        buf = ctypes.create_string_buffer(1 << 18)
        n = L.oracle_icc_make_profile(kind, trc, g, buf, len(buf))
        icc = buf.raw[:n]
        out[name + ".icc"] = np.frombuffer(icc, dtype=np.uint8)
        in8 = rng.integers(0, 256, size=(1, N * 3), dtype=np.uint8)
        in8[0, :30] = np.tile([0, 255, 128], 10)
        o8 = in8.copy()
        assert L.oracle_icc_convert_rows_to_srgb8(icc, n, 0, o8.ctypes.data, N, 1, N * 3) == 0
        out[name + ".in8"], out[name + ".out8"] = in8, o8
        in16 = rng.integers(0, 32769, size=(1, N * 3)).astype(np.uint16)
        in16[0, :30] = np.tile([0, 32768, 16384], 10)
        o16 = in16.copy()
        assert L.oracle_icc_convert_rows_to_srgb16(icc, n, 0, 0, o16.ctypes.data, N, 1, N * 6) == 0
        out[name + ".in16"], out[name + ".out16"] = in16, o16
        if trc in (0, 1, 2):                                  # the 32-bit slices take parametric curves only
            f = np.abs(rng.standard_normal((1, N * 3))).astype(np.float32)
            f[0, :6] = [0, 0, 0, 1, 1, 1]
            a, b = f.copy(), f.copy()
            assert L.oracle_icc_convert_rows_to_rec2020(icc, n, 0, a.ctypes.data, N, 1, N * 12) == 0
            assert L.oracle_icc_convert_rows_to_srgb_float(icc, n, 0, b.ctypes.data, N, 1, N * 12) == 0
            out[name + ".in32"], out[name + ".rec2020"], out[name + ".srgbf"] = f, a, b
    np.savez_compressed(os.path.join(HERE, "icc_vectors.npz"), **out)
    print("written", sum(v.nbytes for v in out.values()), "bytes of vectors")


if __name__ == "__main__":
    main()
