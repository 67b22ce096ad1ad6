"""A third anchor for the oracle's curves (besides the SURVEY known-answer values and the reference line citations): the
published standards themselves, evaluated here in float64 straight from their text -- SMPTE ST 2084 / ITU-R BT.2100 Table 4
(PQ), BT.2100 Table 5 (HLG), SMPTE ST 428-1 (2.6 gamma, 52.37 normalisation), ITU-T H.273 (Kr/Kb, limited-range scaling).
The reference implements these standards in float32; the oracle must agree with the float64 values to float32 accuracy and
hit the landmark numbers every HDR engineer knows."""
import ctypes

import numpy as np
import pytest

import harness

pkg = harness.pkg

M1, M2 = 2610.0 / 16384.0, 2523.0 / 4096.0 * 128.0
C1, C2, C3 = 3424.0 / 4096.0, 2413.0 / 4096.0 * 32.0, 2392.0 / 4096.0 * 32.0


def pq_inverse_eotf(nits):                       # BT.2100 Table 4: E' = ((c1 + c2 Y^m1) / (1 + c3 Y^m1))^m2, Y = F_D / 10000
    y = np.asarray(nits, dtype=np.float64) / 10000.0
    return ((C1 + C2 * y ** M1) / (1.0 + C3 * y ** M1)) ** M2


def pq_eotf(e):                                   # F_D = 10000 * (max(E'^(1/m2) - c1, 0) / (c2 - c3 E'^(1/m2)))^(1/m1)
    p = np.asarray(e, dtype=np.float64) ** (1.0 / M2)
    return 10000.0 * (np.maximum(p - C1, 0.0) / (C2 - C3 * p)) ** (1.0 / M1)


def test_pq_landmarks_and_curve(oracle):
    # landmark code values quoted in BT.2100 / ST 2084 material: 10000 nits -> 1.0, 1000 -> 0.7518, 100 -> 0.5081, 0 -> 0
    for nits, want in ((10000.0, 1.0), (1000.0, 0.751827), (100.0, 0.508078), (0.0, 7.3096e-7)):
        assert abs(float(pq_inverse_eotf(nits)) - want) < 2e-6
    # the plug-in's scale: linear 1.0 = `peak` nits (ColorTransfer.cpp:86: value * peak / 10000)
    for peak in (80.0, 203.0, 1000.0, 10000.0):
        for lin in (0.0, 1e-4, 0.01, 0.18, 1.0, 3.7, 10000.0 / peak):
            if lin * peak > 10000.0:
                continue                              # beyond the curve's domain (the 78.84 exponent amplifies float32 rounding there)
            got = oracle.oracle_linear_to_pq(lin, peak)
            assert abs(got - float(pq_inverse_eotf(lin * peak))) < 3e-6, (peak, lin, got)
    # and back (ColorTransfer.cpp:114: result * 10000 / peak)
    for peak in (80.0, 1000.0):
        for e in (0.05, 0.3, 0.508078, 0.751827, 0.95):
            got = oracle.oracle_pq_to_linear(e, peak)
            want = float(pq_eotf(e)) / peak
            assert abs(got - want) <= 1e-4 * abs(want) + 1e-9, (peak, e, got, want)


def test_hlg_landmarks_and_curve(oracle):
    a = 0.17883277
    b, c = 1.0 - 4.0 * a, 0.5 - a * np.log(4.0 * a)      # BT.2100 Table 5: b = 0.28466892, c = 0.55991073
    assert abs(b - 0.28466892) < 1e-8 and abs(c - 0.55991073) < 1e-8
    def oetf(e):
        return np.sqrt(3.0 * e) if e <= 1.0 / 12.0 else a * np.log(12.0 * e - b) + c
    assert abs(oetf(1.0 / 12.0) - 0.5) < 1e-12 and abs(oetf(1.0) - 1.0) < 1e-7
    for e in (0.0, 0.001, 1.0 / 12.0, 0.0834, 0.25, 0.5, 1.0):
        assert abs(oracle.oracle_linear_to_hlg(e) - oetf(e)) < 2e-6, e
    for v in (0.0, 0.2, 0.5, 0.50001, 0.75, 1.0):        # inverse OETF: E = E'^2 / 3 or (exp((E' - c) / a) + b) / 12
        want = v * v / 3.0 if v <= 0.5 else (np.exp((v - c) / a) + b) / 12.0
        assert abs(oracle.oracle_hlg_to_linear(v) - want) < 2e-6, v


def test_smpte428(oracle):
    # ST 428-1: X' = (X * 48 / 52.37)^(1/2.6) with linear 1.0 = 48 cd/m2 peak white
    for x in (0.0, 0.01, 0.18, 0.5, 1.0):
        want = (x * 48.0 / 52.37) ** (1.0 / 2.6)
        assert abs(oracle.oracle_linear_to_smpte428(x) - want) < 2e-6
    assert abs(oracle.oracle_linear_to_smpte428(1.0) - 0.967043) < 2e-6
    for v in (0.1, 0.5, 0.967043):
        assert abs(oracle.oracle_smpte428_to_linear(v) - (v ** 2.6) * 52.37 / 48.0) < 3e-6


@pytest.mark.parametrize("matrix,kr,kb", [(pkg.MATRIX_BT709, 0.2126, 0.0722), (pkg.MATRIX_BT601, 0.299, 0.114),
                                          (pkg.MATRIX_BT2020_NCL, 0.2627, 0.0593), (pkg.MATRIX_SMPTE240M, 0.212, 0.087),
                                          (pkg.MATRIX_FCC, 0.30, 0.11)])
def test_h273_matrix_coefficients(oracle, matrix, kr, kb):
    out = (ctypes.c_float * 3)()
    oracle.oracle_get_yuv_coefficients(1, matrix, pkg.PRIMARIES_BT709, ctypes.byref(out))
    assert abs(out[0] - kr) < 1e-7 and abs(out[2] - kb) < 1e-7 and abs(out[1] - (1.0 - kr - kb)) < 1e-6


def test_h273_limited_range_endpoints(oracle):
    # H.273 eq. 20-22 video range: Y 16..235 (x 2^(n-8)), C 16..240; black/white/achromatic map to the full-range ends
    for bits in (8, 10, 12):
        s, full = 1 << (bits - 8), (1 << bits) - 1
        assert oracle.oracle_limited_to_full_y(bits, 16 * s) == 0 and oracle.oracle_limited_to_full_y(bits, 235 * s) == full
        assert oracle.oracle_limited_to_full_uv(bits, 16 * s) == 0 and oracle.oracle_limited_to_full_uv(bits, 240 * s) == full
        assert oracle.oracle_limited_to_full_y(bits, 0) == 0 and oracle.oracle_limited_to_full_y(bits, full) == full   # clamped
        mid = oracle.oracle_limited_to_full_uv(bits, 128 * s)
        assert abs(mid - full / 2.0) <= 1.0
