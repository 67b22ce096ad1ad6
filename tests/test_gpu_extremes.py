"""Extreme geometries (Photoshop's PSB limit is 300,000 pixels per side; FormatRecord carries 32-bit coordinates,
AvifFormat.cpp:113-116): very wide and very tall frames against the oracle, both directions, plus a 1-pixel frame.
Index arithmetic (64-bit row offsets, 32-bit group indices, grid-stride launch) is what these exercise."""
import numpy as np
import pytest

import harness

pkg = harness.pkg
pytestmark = pytest.mark.gpu

SHAPES = [(300000, 6), (7, 300000), (299999, 3), (1, 1), (2, 1), (1, 2), (65537, 33)]


@pytest.mark.parametrize("w,h", SHAPES)
def test_write_extreme_shapes(gpu, w, h):
    for kw in (dict(depth=8, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT709),
               dict(depth=16, planes=4, bit_depth=12, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_REFERENCE),
               dict(depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=1000, output=pkg.OUT_YCBCR,
                    chroma=pkg.CHROMA_422, matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020),
               # round 4's streaming kernels: any width on RGB f32 4:4:4, and the transparent document's default save (RGBA f32 -> 4:2:0 + alpha)
               dict(depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=80, output=pkg.OUT_YCBCR,
                    chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020),
               dict(depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=80, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR,
                    chroma=pkg.CHROMA_420, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                    color_primaries=pkg.PRIMARIES_BT2020)):
        d = pkg.WriteDesc(width=w, height=h, **kw)
        src = harness.make_write_source(d, seed=w % 97)
        want = harness.oracle_write(d, src)
        got = harness.gpu_write(gpu, d, src, mem="device")
        st = harness.compare_write(d, want, got)
        if d.depth == 32:
            assert st["max_abs"] <= 1 and harness.t2_exact_ok(st, 0.998), (w, h, st)
        else:
            assert st["max_abs"] == 0, (w, h, kw, st)


@pytest.mark.parametrize("w,h", SHAPES)
def test_read_extreme_shapes(gpu, w, h):
    for kw in (dict(colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_420, bit_depth=8, depth=8, alpha_state=pkg.ALPHA_NONE),
               dict(colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_422, bit_depth=12, depth=16, alpha_state=pkg.ALPHA_PREMULTIPLIED,
                    matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020),
               dict(colorspace=pkg.COLORSPACE_MONOCHROME, chroma=pkg.CHROMA_MONOCHROME, bit_depth=10, depth=16, alpha_state=pkg.ALPHA_STRAIGHT)):
        d = pkg.ReadDesc(width=w, height=h, **kw)
        planes = harness.make_read_source(d, seed=h % 89)
        want = harness.oracle_read(d, planes)
        got = harness.gpu_read(gpu, d, planes, mem="device")
        assert np.array_equal(got, want), (w, h, kw)
    # an HDR open (f32 host, tier 2): what the default HDR save decodes to
    d = pkg.ReadDesc(width=w, height=h, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_422, bit_depth=12, depth=32, alpha_state=pkg.ALPHA_NONE,
                     transfer_characteristics=pkg.TC_PQ, pq_peak_nits=80, matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
    planes = harness.make_read_source(d, seed=h % 89)
    w64 = harness.oracle_read(d, planes).astype(np.float64)
    g64 = harness.gpu_read(gpu, d, planes, mem="device").astype(np.float64)
    assert np.all(np.isfinite(g64)) and np.all(np.abs(g64 - w64) <= 1e-4 * np.abs(w64) + 1e-9), (w, h)          # the T2 bar of tests/test_gpu_read.py


@pytest.mark.parametrize("transfer", [pkg.TRANSFER_PQ, pkg.TRANSFER_HLG, pkg.TRANSFER_SMPTE428, pkg.TRANSFER_CLIP])
@pytest.mark.parametrize("width,output", [(512, pkg.OUT_YCBCR), (67, pkg.OUT_YCBCR), (512, pkg.OUT_REFERENCE)])
def test_non_finite_samples_stay_in_range(gpu, transfer, width, output):
    """NaN and +-Inf samples are outside the parity contract (the reference casts NaN to uint16_t: undefined behaviour,
    WriteHeifImage.cpp:1093).  What the library guarantees instead: no crash, every code inside [0, max], NaN (quiet or
    signalling) and -Inf -> 0, +Inf -> 0 or the maximum code (inf/inf inside the PQ curve is a NaN, the other curves saturate),
    and finite neighbours unaffected (streaming and generic kernels alike)."""
    d = pkg.WriteDesc(width=width, height=4, depth=32, planes=3, bit_depth=10, transfer=transfer, peak_nits=80,
                      alpha_state=pkg.ALPHA_NONE, output=output, chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_RGB_GBR)
    src = harness.make_write_source(d, seed=1)
    clean = harness.gpu_write(gpu, d, src)
    bad = src.copy().reshape(4, width, 3)
    bad[0, 5, :] = np.nan
    bad[1, 6, :] = np.inf
    bad[2, 7, :] = -np.inf
    bad[3, 8, 1] = np.array([0x7fa00000], dtype=np.uint32).view(np.float32)[0]          # a signalling NaN bit pattern
    got = harness.gpu_write(gpu, d, bad.reshape(4, -1))
    if output == pkg.OUT_YCBCR:                                # identity matrix: Y <- G, Cb <- B, Cr <- R, so planes are the codes
        g = np.stack([got[2], got[0], got[1]], axis=-1)        # back to R, G, B order
        c = np.stack([clean[2], clean[0], clean[1]], axis=-1)
    else:
        g, c = got[0].reshape(4, width, 3), clean[0].reshape(4, width, 3)
    assert g.max() <= 1023
    assert np.all(g[0, 5] == 0) and np.all(g[2, 7] == 0) and g[3, 8, 1] == 0
    assert np.all((g[1, 6] == 1023) | (g[1, 6] == 0))
    mask = np.ones((4, width), bool)
    mask[0, 5] = mask[1, 6] = mask[2, 7] = mask[3, 8] = False
    assert np.array_equal(g[mask], c[mask])
