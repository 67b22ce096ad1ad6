"""Every specialised streaming kernel must produce the SAME BYTES as the generic write_px kernel on the same input (they share
the device functions, so this is a check of index arithmetic, LDS transposes and edge handling, not of numerics).
avifgpu_set_hot_variant(0) switches all of them off; the default tuning word switches them on."""
import numpy as np
import pytest

import harness

pkg = harness.pkg
pytestmark = pytest.mark.gpu

BT2020 = dict(matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
CASES = [
    ("write_rgb32_ycbcr444_hot", dict(width=1024, height=9, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=203,
                                      output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020)),
    ("write_rgb32_ycbcr_sub_hot", dict(width=1536, height=7, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=1000,
                                       output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, **BT2020)),
    ("write_rgb32_ycbcr_sub_hot", dict(width=512, height=5, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_SMPTE428,
                                       output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422, **BT2020)),
    ("write_rgba32_ycbcra444_hot", dict(width=768, height=6, depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                                        alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020)),
    # widths that are not whole spans: the masked last span of every row (real document geometries, e.g. 7952 x 5304)
    ("write_rgb32_ycbcr444_hot", dict(width=1004, height=9, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                                      output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020)),
    ("write_rgb32_ycbcr444_hot", dict(width=36, height=3, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_CLIP,
                                      output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020)),
    ("write_rgb32_ycbcr_sub_hot", dict(width=1500, height=7, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=1000,
                                       output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, **BT2020)),
    ("write_rgb32_ycbcr_sub_hot", dict(width=1500, height=6, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=1000,
                                       output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, **BT2020)),
    ("write_rgb32_ycbcr_sub_hot", dict(width=524, height=5, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_HLG,
                                       output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422, **BT2020)),
    ("write_rgba32_ycbcra444_hot", dict(width=771, height=6, depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                                        alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020)),
    ("write_f32_ref_stream", dict(width=1000, height=5, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                                  output=pkg.OUT_REFERENCE)),
    ("write_f32_ref_stream", dict(width=333, height=4, depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_CLIP,
                                  alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_REFERENCE)),
    ("write_rgb16_ycbcr444_hot", dict(width=1024, height=5, depth=16, planes=3, bit_depth=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020)),
    ("write_rgb16_ycbcr444_hot", dict(width=1000, height=4, depth=16, planes=3, bit_depth=10, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                                      matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb16_ycbcr444_hot", dict(width=8, height=3, depth=16, planes=3, bit_depth=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                                      matrix_coefficients=pkg.MATRIX_RGB_GBR)),
    ("write_rgb16_ycbcr_sub_hot", dict(width=1024, height=7, depth=16, planes=3, bit_depth=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, **BT2020)),
    ("write_rgb16_ycbcr_sub_hot", dict(width=1000, height=6, depth=16, planes=3, bit_depth=10, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                                       chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb16_ycbcr_sub_hot", dict(width=520, height=5, depth=16, planes=3, bit_depth=10, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                       matrix_coefficients=pkg.MATRIX_BT709, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb16_ycbcr_sub_hot", dict(width=8, height=1, depth=16, planes=3, bit_depth=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, **BT2020)),
    ("write_rgba16_ycbcra444_hot", dict(width=1024, height=5, depth=16, planes=4, bit_depth=12, alpha_state=pkg.ALPHA_PREMULTIPLIED,
                                        output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020)),
    ("write_rgba16_ycbcra444_hot", dict(width=1000, height=4, depth=16, planes=4, bit_depth=10, alpha_state=pkg.ALPHA_STRAIGHT,
                                        output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgba16_ycbcra444_hot", dict(width=8, height=2, depth=16, planes=4, bit_depth=12, alpha_state=pkg.ALPHA_PREMULTIPLIED,
                                        output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020)),
    ("write_int_ref_stream", dict(width=1000, height=5, depth=16, planes=3, bit_depth=12, output=pkg.OUT_REFERENCE)),
    ("write_int_ref_stream", dict(width=502, height=3, depth=16, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED,
                                  output=pkg.OUT_REFERENCE)),
]


@pytest.mark.parametrize("kernel,kw", CASES, ids=[f"{k}-{i}" for i, (k, _) in enumerate(CASES)])
def test_specialised_kernel_equals_generic(gpu, kernel, kw):
    d = pkg.WriteDesc(**kw)
    src = harness.make_write_source(d, seed=31)
    try:
        gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4 | 8)          # bit 3: streaming kernels that the default only takes for large frames
        fast = harness.gpu_write(gpu, d, src, mem="device")
        assert kernel in gpu.last_kernel(), gpu.last_kernel()
        gpu.lib.avifgpu_set_hot_variant(0)
        slow = harness.gpu_write(gpu, d, src, mem="device")
        assert "write_px" in gpu.last_kernel(), gpu.last_kernel()
    finally:
        gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4)
    for pl in slow:
        assert np.array_equal(fast[pl], slow[pl]), (kernel, pl)
