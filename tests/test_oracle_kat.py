"""Pins the CPU oracle to the known-answer values SURVEY.md section 8(c) recorded from the reference's own code.

These are the only reference-derived vectors that exist for this path (the reference has no tests and cannot be
built in this image -- DESIGN.md "Oracle"), so every one of them is asserted, to the 9 significant digits printed.
"""
import ctypes
import json
import os

import numpy as np

import harness

pkg = harness.pkg
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_kat.json")))


def _g9(x):
    return float("%.9g" % x)


def test_scalar_curves(oracle):
    for name, args, want in KAT["scalar"]:
        got = getattr(oracle, name)(*args)
        assert _g9(got) == want, (name, args, got, want)


def test_premultiply(oracle):
    for c, a, want in KAT["premultiply_u8"]:
        assert oracle.oracle_premultiply_u8(c, a) == want
    for c, a, m, want in KAT["premultiply_u16"]:
        assert oracle.oracle_premultiply_u16(c, a, m) == want
    for c, a, want in KAT["unpremultiply_u8"]:
        assert oracle.oracle_unpremultiply_u8(c, a) == want


def test_bt2020_coefficients(oracle):
    out = (ctypes.c_float * 3)()
    oracle.oracle_get_yuv_coefficients(1, pkg.MATRIX_BT2020_NCL, pkg.PRIMARIES_BT2020, ctypes.byref(out))
    assert [_g9(v) for v in out] == KAT["bt2020_ncl_kr_kg_kb"]


def test_write_pixel():
    k = KAT["write_pixel_pq80_10bit"]
    d = pkg.WriteDesc(width=1, height=1, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                      alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
    src = np.array([k["rgb"]], dtype=np.float32)
    out = harness.oracle_write(d, src)
    assert out[0][0].tolist() == k["codes"]


def test_read_pixel_10bit_pq():
    k = KAT["read_pixel_10bit_bt2020_pq80"]
    d = pkg.ReadDesc(width=1, height=1, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_444, bit_depth=10, depth=32,
                     alpha_state=pkg.ALPHA_NONE, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                     color_primaries=pkg.PRIMARIES_BT2020, transfer_characteristics=pkg.TC_PQ, pq_peak_nits=80)
    planes = {i: np.array([[k["yuv"][i]] + [0] * 7], dtype=np.uint16) for i in range(3)}
    out = harness.oracle_read(d, planes)
    assert [_g9(v) for v in out[0]] == k["rgb"]


def test_read_pixel_8bit_709():
    k = KAT["read_pixel_8bit_bt709"]
    d = pkg.ReadDesc(width=1, height=1, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_444, bit_depth=8, depth=8,
                     alpha_state=pkg.ALPHA_NONE, matrix_coefficients=pkg.MATRIX_BT709)
    planes = {i: np.array([[k["yuv"][i]] + [0] * 7], dtype=np.uint8) for i in range(3)}
    out = harness.oracle_read(d, planes)
    assert out[0].tolist() == k["rgb"]
