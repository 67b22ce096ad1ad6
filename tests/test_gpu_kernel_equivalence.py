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
    # round 5: RGB8 -> u8 planes, the plug-in's default save (whole spans of 1024 pixels, a ragged last span, a single 8-pixel group; odd heights)
    ("write_rgb8_ycbcr_hot", dict(width=2048, height=6, depth=8, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                                  matrix_coefficients=pkg.MATRIX_BT709, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb8_ycbcr_hot", dict(width=1512, height=7, depth=8, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                                  matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb8_ycbcr_hot", dict(width=1512, height=5, depth=8, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                                  chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb8_ycbcr_hot", dict(width=1000, height=5, depth=8, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                  chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb8_ycbcr_hot", dict(width=1000, height=4, depth=8, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                  matrix_coefficients=pkg.MATRIX_BT709, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb8_ycbcr_hot", dict(width=1032, height=3, depth=8, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020)),
    ("write_rgb8_ycbcr_hot", dict(width=8, height=1, depth=8, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                                  matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb8_ycbcr_hot", dict(width=16, height=2, depth=8, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                                  matrix_coefficients=pkg.MATRIX_RGB_GBR)),
    # round 6: RGB8 -> u16 planes (an 8-bit document saved at 10 / 12 bit): whole spans, a ragged last span inside the first and inside the second
    # 512-pixel half, a single 8-pixel group; odd heights; box and nearest
    ("write_rgb8_ycbcr16_hot", dict(width=2048, height=6, depth=8, planes=3, bit_depth=10, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                                    matrix_coefficients=pkg.MATRIX_BT709, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb8_ycbcr16_hot", dict(width=1512, height=7, depth=8, planes=3, bit_depth=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                                    matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb8_ycbcr16_hot", dict(width=1320, height=5, depth=8, planes=3, bit_depth=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                                    chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb8_ycbcr16_hot", dict(width=1000, height=5, depth=8, planes=3, bit_depth=10, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                    chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb8_ycbcr16_hot", dict(width=1544, height=4, depth=8, planes=3, bit_depth=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                    matrix_coefficients=pkg.MATRIX_BT709, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb8_ycbcr16_hot", dict(width=1032, height=3, depth=8, planes=3, bit_depth=10, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020)),
    ("write_rgb8_ycbcr16_hot", dict(width=8, height=1, depth=8, planes=3, bit_depth=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                                    matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb8_ycbcr16_hot", dict(width=520, height=2, depth=8, planes=3, bit_depth=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                                    matrix_coefficients=pkg.MATRIX_RGB_GBR)),
    # round 5: RGBA8 -> u8 planes + alpha (straight and premultiplied), every chroma format
    ("write_rgba8_ycbcra_hot", dict(width=1024, height=6, depth=8, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR,
                                    chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgba8_ycbcra_hot", dict(width=1000, height=7, depth=8, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR,
                                    chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT709, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgba8_ycbcra_hot", dict(width=1000, height=5, depth=8, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR,
                                    chroma=pkg.CHROMA_420, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgba8_ycbcra_hot", dict(width=520, height=5, depth=8, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR,
                                    chroma=pkg.CHROMA_422, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgba8_ycbcra_hot", dict(width=520, height=4, depth=8, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR,
                                    chroma=pkg.CHROMA_422, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, matrix_coefficients=pkg.MATRIX_BT709, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgba8_ycbcra_hot", dict(width=1032, height=3, depth=8, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR,
                                    chroma=pkg.CHROMA_444, **BT2020)),
    ("write_rgba8_ycbcra_hot", dict(width=8, height=1, depth=8, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR,
                                    chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    # round 5: 16-bit documents saved at 8 bit on the RGB16 streaming kernels (u8 planes)
    ("write_rgb16_ycbcr_sub_hot", dict(width=1024, height=7, depth=16, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                                       matrix_coefficients=pkg.MATRIX_BT709, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb16_ycbcr_sub_hot", dict(width=1000, height=6, depth=16, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                                       chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb16_ycbcr_sub_hot", dict(width=520, height=5, depth=16, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                       matrix_coefficients=pkg.MATRIX_BT709, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb16_ycbcr444_hot", dict(width=1000, height=4, depth=16, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                                      matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgb16_ycbcr444_hot", dict(width=8, height=3, depth=16, planes=3, bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                                      matrix_coefficients=pkg.MATRIX_RGB_GBR)),
    ("write_int_ref_stream", dict(width=1000, height=5, depth=16, planes=3, bit_depth=12, output=pkg.OUT_REFERENCE)),
    ("write_int_ref_stream", dict(width=502, height=3, depth=16, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED,
                                  output=pkg.OUT_REFERENCE)),
    # round 5 (last series): gray documents without alpha ARE their Y plane sample for sample -- the elementwise kernels with one sample per pixel
    ("write_f32_ref_stream", dict(width=1000, height=5, depth=32, planes=1, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80)),
    ("write_f32_ref_stream", dict(width=1028, height=3, depth=32, planes=1, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=1000)),
    ("write_f32_ref_stream", dict(width=516, height=4, depth=32, planes=1, bit_depth=12, transfer=pkg.TRANSFER_CLIP)),       # (gray saves take PQ or Clip only, WriteHeifImage.cpp:581)
    ("write_f32_ref_stream", dict(width=4, height=2, depth=32, planes=1, bit_depth=10, transfer=pkg.TRANSFER_CLIP)),
    ("write_int_ref_stream", dict(width=1000, height=5, depth=16, planes=1, bit_depth=12)),
    ("write_int_ref_stream", dict(width=1032, height=3, depth=16, planes=1, bit_depth=10)),
    ("write_int_ref_stream", dict(width=8, height=2, depth=16, planes=1, bit_depth=8)),
    # ... a transparent 16-bit document saved 4:2:2 / 4:2:0 (the plug-in's default is 4:2:2): box and nearest, premultiplied and straight, odd heights
    ("write_rgba16_ycbcra_sub_hot", dict(width=1024, height=6, depth=16, planes=4, bit_depth=12, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR,
                                         chroma=pkg.CHROMA_422, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, **BT2020)),
    ("write_rgba16_ycbcra_sub_hot", dict(width=1000, height=7, depth=16, planes=4, bit_depth=10, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR,
                                         chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT709, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgba16_ycbcra_sub_hot", dict(width=520, height=5, depth=16, planes=4, bit_depth=12, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR,
                                         chroma=pkg.CHROMA_420, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgba16_ycbcra_sub_hot", dict(width=1512, height=4, depth=16, planes=4, bit_depth=10, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR,
                                         chroma=pkg.CHROMA_422, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgba16_ycbcra_sub_hot", dict(width=8, height=1, depth=16, planes=4, bit_depth=12, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR,
                                         chroma=pkg.CHROMA_420, **BT2020)),
    # round 6: ... saved at 8 bit (u8 planes): 4:2:0 / 4:2:2 / 4:4:4, premultiplied and straight
    ("write_rgba16_ycbcra_sub_hot", dict(width=1024, height=6, depth=16, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR,
                                         chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgba16_ycbcra_sub_hot", dict(width=1000, height=7, depth=16, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR,
                                         chroma=pkg.CHROMA_420, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, matrix_coefficients=pkg.MATRIX_BT709, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgba16_ycbcra_sub_hot", dict(width=520, height=5, depth=16, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR,
                                         chroma=pkg.CHROMA_422, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgba16_ycbcra_sub_hot", dict(width=8, height=1, depth=16, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR,
                                         chroma=pkg.CHROMA_422, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    ("write_rgba16_ycbcra444_hot", dict(width=1000, height=4, depth=16, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED,
                                        output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)),
    # round 6: 8-bit documents through the 8-bit hand-off without premultiplication are a copy (contiguous rows: one flat row; padded rows; ragged ends)
    ("write_copy_rows_stream", dict(width=1024, height=5, depth=8, planes=3, bit_depth=8, output=pkg.OUT_REFERENCE)),
    ("write_copy_rows_stream", dict(width=1000, height=7, depth=8, planes=3, bit_depth=8, output=pkg.OUT_REFERENCE)),
    ("write_copy_rows_stream", dict(width=2732, height=3, depth=8, planes=3, bit_depth=8, output=pkg.OUT_REFERENCE)),
    ("write_copy_rows_stream", dict(width=1001, height=4, depth=8, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_REFERENCE)),
    ("write_copy_rows_stream", dict(width=4100, height=3, depth=8, planes=1, bit_depth=8)),
    ("write_copy_rows_stream", dict(width=4, height=1, depth=8, planes=3, bit_depth=8, output=pkg.OUT_REFERENCE)),
    # ... and gray + alpha: two interleaved samples per pixel, two planes out, stage_a itself per pixel (write_ga_stream)
    ("write_ga_stream", dict(width=1000, height=5, depth=16, planes=2, bit_depth=12, alpha_state=pkg.ALPHA_PREMULTIPLIED)),
    ("write_ga_stream", dict(width=1028, height=3, depth=16, planes=2, bit_depth=10, alpha_state=pkg.ALPHA_STRAIGHT)),
    ("write_ga_stream", dict(width=4, height=2, depth=16, planes=2, bit_depth=12, alpha_state=pkg.ALPHA_PREMULTIPLIED)),
    ("write_ga_stream", dict(width=1000, height=5, depth=32, planes=2, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80, alpha_state=pkg.ALPHA_PREMULTIPLIED)),
    ("write_ga_stream", dict(width=514, height=3, depth=32, planes=2, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=1000, alpha_state=pkg.ALPHA_STRAIGHT)),
    ("write_ga_stream", dict(width=2, height=2, depth=32, planes=2, bit_depth=12, transfer=pkg.TRANSFER_CLIP, alpha_state=pkg.ALPHA_PREMULTIPLIED)),
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


FLAT_CASES = [
    dict(width=1000, height=7, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020),
    dict(width=7952, height=3, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=1000, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020),
    dict(width=1016, height=5, depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=80, alpha_state=pkg.ALPHA_STRAIGHT,
         output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020),
    dict(width=1000, height=5, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80, output=pkg.OUT_REFERENCE),
    dict(width=1000, height=6, depth=16, planes=3, bit_depth=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020),
    dict(width=1000, height=6, depth=16, planes=4, bit_depth=10, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020),
    dict(width=1000, height=5, depth=16, planes=3, bit_depth=12, output=pkg.OUT_REFERENCE),
    dict(width=16, height=9, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_SMPTE428, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, **BT2020),
    # 4:2:2 is sub-sampled along the row only: flat too (chroma planes of width / 2)
    dict(width=1008, height=7, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422, **BT2020),
    dict(width=1008, height=6, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=1000, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
         chroma_downsampling=pkg.DOWNSAMPLE_NEAREST, **BT2020),
    dict(width=1008, height=5, depth=16, planes=3, bit_depth=10, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422, matrix_coefficients=pkg.MATRIX_BT709),
    dict(width=1008, height=5, depth=16, planes=4, bit_depth=12, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422, **BT2020),
]


@pytest.mark.parametrize("kw", FLAT_CASES, ids=[f"flat-{i}" for i in range(len(FLAT_CASES))])
def test_flat_launch_equals_row_launch(gpu, kw):
    """launch_write: a contiguous 4:4:4 / interleaved tile is launched as ONE row of width x nrows pixels (span boundaries of the
    buffer instead of row boundaries).  Same kernels, same arithmetic: the bytes must equal the row-by-row launch (variant bit 4 = off)
    and the generic kernel's."""
    d = pkg.WriteDesc(**kw)
    src = harness.make_write_source(d, seed=17)
    try:
        gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4 | 8)
        flat = harness.gpu_write(gpu, d, src, mem="device")
        assert gpu.last_kernel().endswith(" flat"), gpu.last_kernel()
        gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4 | 8 | 16)
        rows = harness.gpu_write(gpu, d, src, mem="device")
        assert "flat" not in gpu.last_kernel(), gpu.last_kernel()
        gpu.lib.avifgpu_set_hot_variant(0)
        slow = harness.gpu_write(gpu, d, src, mem="device")
        assert "write_px" in gpu.last_kernel() and "flat" not in gpu.last_kernel(), gpu.last_kernel()
    finally:
        gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4)
    for pl in slow:
        assert np.array_equal(flat[pl], rows[pl]), pl
        assert np.array_equal(flat[pl], slow[pl]), pl
    # a padded row stride (what libheif hands over for widths that are not a multiple of 8 samples) is NOT flattened
    harness.gpu_write(gpu, d, src, mem="device", stride_pad=8)
    assert "flat" not in gpu.last_kernel(), gpu.last_kernel()


READ_FLAT_CASES = [
    dict(width=1000, height=7, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_444, bit_depth=10, depth=32, alpha_state=pkg.ALPHA_NONE,
         matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020, transfer_characteristics=pkg.TC_PQ),
    dict(width=1000, height=6, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_444, bit_depth=12, depth=16, alpha_state=pkg.ALPHA_PREMULTIPLIED,
         matrix_coefficients=pkg.MATRIX_BT709),
    dict(width=1008, height=5, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_444, bit_depth=8, depth=8, alpha_state=pkg.ALPHA_NONE,
         matrix_coefficients=pkg.MATRIX_BT601, full_range_flag=0),
    dict(width=1008, height=9, colorspace=pkg.COLORSPACE_MONOCHROME, chroma=pkg.CHROMA_MONOCHROME, bit_depth=8, depth=8, alpha_state=pkg.ALPHA_STRAIGHT),
    dict(width=1000, height=4, colorspace=pkg.COLORSPACE_MONOCHROME, chroma=pkg.CHROMA_MONOCHROME, bit_depth=12, depth=16, alpha_state=pkg.ALPHA_NONE),
    dict(width=1000, height=5, colorspace=pkg.COLORSPACE_RGB, chroma=pkg.CHROMA_444, bit_depth=12, depth=32, alpha_state=pkg.ALPHA_STRAIGHT,
         matrix_coefficients=pkg.MATRIX_RGB_GBR, color_primaries=pkg.PRIMARIES_BT2020, transfer_characteristics=pkg.TC_HLG),
    dict(width=1008, height=6, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_422, bit_depth=8, depth=8, alpha_state=pkg.ALPHA_NONE,
         matrix_coefficients=pkg.MATRIX_BT601),
    dict(width=1008, height=5, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_422, bit_depth=10, depth=32, alpha_state=pkg.ALPHA_STRAIGHT,
         matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020, transfer_characteristics=pkg.TC_PQ),
]


@pytest.mark.parametrize("kw", READ_FLAT_CASES, ids=[f"read-flat-{i}" for i in range(len(READ_FLAT_CASES))])
def test_flat_read_launch_equals_row_launch(gpu, kw):
    """launch_read: planes without chroma sub-sampling and a host row buffer that are all contiguous are decoded as ONE row of
    width x nrows pixels.  Same kernel: the bytes must equal the row-by-row launch (variant bit 4 = off) -- and the oracle."""
    d = pkg.ReadDesc(**kw)
    planes = harness.make_read_source(d, seed=23)
    try:
        flat = harness.gpu_read(gpu, d, planes)
        assert gpu.last_kernel().endswith(" flat"), gpu.last_kernel()
        gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4 | 16)
        rows = harness.gpu_read(gpu, d, planes)
        assert "flat" not in gpu.last_kernel(), gpu.last_kernel()
    finally:
        gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4)
    assert np.array_equal(flat.view(np.uint8), rows.view(np.uint8))
    want = harness.oracle_read(d, planes)
    if d.depth == 32:
        np.testing.assert_allclose(flat, want, rtol=1e-4, atol=1e-9)
    else:
        assert np.array_equal(flat, want)
    harness.gpu_read(gpu, d, harness.make_read_source(d, seed=23, stride_pad=8))     # padded plane strides: not flattened
    assert "flat" not in gpu.last_kernel(), gpu.last_kernel()
