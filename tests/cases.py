"""Parity case tables shared by the CPU (oracle/golden) tests and the GPU parity tests.

Sizes are deliberately awkward (odd widths/heights, not multiples of the 4/8-pixel thread footprint) so the
edge paths run; BASELINE.json's configs appear at reduced size here and at full size in test_gpu_fullsize.py.
"""
from __future__ import annotations

import harness

pkg = harness.pkg

W_ODD, H_ODD = 67, 21        # ragged right edge + odd height
W_EVEN, H_EVEN = 96, 16


def _alpha_states(planes):
    return [pkg.ALPHA_STRAIGHT, pkg.ALPHA_PREMULTIPLIED] if planes in (2, 4) else [pkg.ALPHA_NONE]


def write_cases():
    """[(id, kwargs)] -- every CreateHeifImage* branch (reference WriteHeifImage.cpp:169-1139) + fused stage B."""
    out = []
    # ---- integer sources, reference output (bit-exact tier) ----
    for depth in (8, 16):
        for planes in (1, 2, 3, 4):
            for bits in (8, 10, 12):
                for a in _alpha_states(planes):
                    out.append((f"ref-d{depth}-p{planes}-b{bits}-a{a}",
                                dict(width=W_ODD, height=H_ODD, depth=depth, planes=planes, bit_depth=bits,
                                     alpha_state=a, output=pkg.OUT_REFERENCE)))
    # ---- float sources, reference output ----
    for planes in (1, 2):
        for tr in (pkg.TRANSFER_PQ, pkg.TRANSFER_CLIP):
            for bits in (10, 12):
                for a in _alpha_states(planes):
                    out.append((f"ref-d32-p{planes}-b{bits}-t{tr}-a{a}",
                                dict(width=W_ODD, height=H_ODD, depth=32, planes=planes, bit_depth=bits, transfer=tr,
                                     peak_nits=80, alpha_state=a, output=pkg.OUT_REFERENCE)))
    for planes in (3, 4):
        for tr, bits_list in ((pkg.TRANSFER_PQ, (10, 12)), (pkg.TRANSFER_SMPTE428, (12,)),
                              (pkg.TRANSFER_CLIP, (10, 12)), (pkg.TRANSFER_HLG, (10,))):
            for bits in bits_list:
                for a in _alpha_states(planes):
                    out.append((f"ref-d32-p{planes}-b{bits}-t{tr}-a{a}",
                                dict(width=W_ODD, height=H_ODD, depth=32, planes=planes, bit_depth=bits, transfer=tr,
                                     peak_nits=80, alpha_state=a, output=pkg.OUT_REFERENCE)))
    for peak in (1, 1000, 10000):
        out.append((f"ref-d32-p3-b12-pq{peak}",
                    dict(width=W_EVEN, height=H_EVEN, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_PQ,
                         peak_nits=peak, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)))
    # ---- fused YCbCr output ----
    mats = [(pkg.MATRIX_BT601, pkg.PRIMARIES_BT709), (pkg.MATRIX_BT709, pkg.PRIMARIES_BT709),
            (pkg.MATRIX_BT2020_NCL, pkg.PRIMARIES_BT2020)]
    for depth, bits_list in ((8, (8, 10)), (16, (8, 12)), (32, (10, 12))):
        for planes in (3, 4):
            for chroma in (pkg.CHROMA_444, pkg.CHROMA_422, pkg.CHROMA_420):
                for bits in bits_list:
                    for (m, pr) in mats:
                        a = pkg.ALPHA_STRAIGHT if planes == 4 else pkg.ALPHA_NONE
                        kw = dict(width=W_ODD, height=H_ODD, depth=depth, planes=planes, bit_depth=bits, alpha_state=a,
                                  output=pkg.OUT_YCBCR, chroma=chroma, matrix_coefficients=m, color_primaries=pr)
                        if depth == 32:
                            kw.update(transfer=pkg.TRANSFER_PQ, peak_nits=80)
                        out.append((f"ycc-d{depth}-p{planes}-b{bits}-c{chroma}-m{m}", kw))
                        if chroma != pkg.CHROMA_444:
                            # libheif 1.14.0's own down-sampling (top-left sample) = the FormatRecord shim's default
                            out.append((f"ycc-d{depth}-p{planes}-b{bits}-c{chroma}-m{m}-near",
                                        dict(kw, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST)))
    extra = [
        ("ycc-d8-p3-b8-420-nearest", dict(width=W_ODD, height=H_ODD, depth=8, planes=3, bit_depth=8,
                                          alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                                          matrix_coefficients=pkg.MATRIX_BT709, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST)),
        ("ycc-d8-p4-b8-444-gbr-premul", dict(width=W_ODD, height=H_ODD, depth=8, planes=4, bit_depth=8,
                                             alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_YCBCR,
                                             chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_RGB_GBR)),
        ("ycc-d16-p3-b12-444-chromaderived", dict(width=W_EVEN, height=H_EVEN, depth=16, planes=3, bit_depth=12,
                                                  alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                                                  matrix_coefficients=pkg.MATRIX_CHROMA_DERIVED_NCL,
                                                  color_primaries=pkg.PRIMARIES_SMPTE432)),
        ("ycc-d32-p3-b12-422-428", dict(width=W_ODD, height=H_ODD, depth=32, planes=3, bit_depth=12,
                                        transfer=pkg.TRANSFER_SMPTE428, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                                        chroma=pkg.CHROMA_422, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                        color_primaries=pkg.PRIMARIES_BT2020)),
        ("ycc-d32-p4-b10-444-clip-premul", dict(width=W_ODD, height=H_ODD, depth=32, planes=4, bit_depth=10,
                                                transfer=pkg.TRANSFER_CLIP, alpha_state=pkg.ALPHA_PREMULTIPLIED,
                                                output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                                                matrix_coefficients=pkg.MATRIX_BT601)),
        # the LDS-transposed hot kernel needs width % 256 == 0
        ("ycc-d32-p3-b10-444-hot", dict(width=512, height=6, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ,
                                        peak_nits=80, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                                        chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                        color_primaries=pkg.PRIMARIES_BT2020)),
        ("ycc-d32-p3-b12-444-hot1000", dict(width=256, height=5, depth=32, planes=3, bit_depth=12,
                                            transfer=pkg.TRANSFER_PQ, peak_nits=1000, alpha_state=pkg.ALPHA_NONE,
                                            output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                                            matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                            color_primaries=pkg.PRIMARIES_BT2020)),
        # ... and its 4:2:0 / 4:2:2 sibling width % 512 == 0 (odd heights: the last chroma row replicates the image edge)
        ("ycc-d32-p3-b10-420-hot", dict(width=1024, height=7, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ,
                                        peak_nits=80, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                                        chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                        color_primaries=pkg.PRIMARIES_BT2020)),
        ("ycc-d32-p3-b12-422-hot428", dict(width=512, height=5, depth=32, planes=3, bit_depth=12,
                                           transfer=pkg.TRANSFER_SMPTE428, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                                           chroma=pkg.CHROMA_422, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                           color_primaries=pkg.PRIMARIES_BT2020)),
        ("ycc-d32-p3-b10-420-hot-nearest-clip", dict(width=512, height=4, depth=32, planes=3, bit_depth=10,
                                                     transfer=pkg.TRANSFER_CLIP, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                                                     chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT601,
                                                     chroma_downsampling=pkg.DOWNSAMPLE_NEAREST)),
        ("ycc-d32-p3-b10-422-hot-hlg", dict(width=1536, height=3, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_HLG,
                                            alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                            matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)),
        # ... and the RGBA f32 -> Y, Cb, Cr, A streaming kernel (width % 256 == 0): straight + premultiplied alpha
        ("ycc-d32-p4-b12-444-hot-straight", dict(width=512, height=5, depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_PQ,
                                                 peak_nits=1000, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR,
                                                 chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                                 color_primaries=pkg.PRIMARIES_BT2020)),
        ("ycc-d32-p4-b10-444-hot-premul-clip", dict(width=256, height=6, depth=32, planes=4, bit_depth=10,
                                                    transfer=pkg.TRANSFER_CLIP, alpha_state=pkg.ALPHA_PREMULTIPLIED,
                                                    output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                                                    matrix_coefficients=pkg.MATRIX_BT601)),
        ("ycc-d32-p4-b10-444-hot-gbr-428", dict(width=768, height=3, depth=32, planes=4, bit_depth=10,
                                                transfer=pkg.TRANSFER_SMPTE428, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR,
                                                chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_RGB_GBR)),
        # the streaming kernels on widths that are not whole spans (document sizes like 7952 x 5304): masked last span of each row
        ("ycc-d32-p3-b10-444-hot-tail520", dict(width=520, height=5, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ,
                                                peak_nits=80, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                                                chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                                color_primaries=pkg.PRIMARIES_BT2020)),
        ("ycc-d32-p3-b12-444-hot-tail252", dict(width=252, height=4, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_CLIP,
                                                alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                                                matrix_coefficients=pkg.MATRIX_BT601)),
        ("ycc-d32-p3-b10-420-hot-tail1004", dict(width=1004, height=7, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ,
                                                 peak_nits=1000, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                                                 chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                                 color_primaries=pkg.PRIMARIES_BT2020)),
        ("ycc-d32-p3-b10-420-hot-tail516-near", dict(width=516, height=6, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ,
                                                     peak_nits=80, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                                                     chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                                     color_primaries=pkg.PRIMARIES_BT2020, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST)),
        ("ycc-d32-p3-b12-422-hot-tail260", dict(width=260, height=3, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_SMPTE428,
                                                alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                                matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)),
        ("ycc-d32-p4-b12-444-hot-tail259", dict(width=259, height=5, depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_PQ,
                                                peak_nits=1000, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR,
                                                chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                                color_primaries=pkg.PRIMARIES_BT2020)),
        ("ycc-d32-p4-b10-444-hot-tail513-premul", dict(width=513, height=3, depth=32, planes=4, bit_depth=10,
                                                       transfer=pkg.TRANSFER_CLIP, alpha_state=pkg.ALPHA_PREMULTIPLIED,
                                                       output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                                                       matrix_coefficients=pkg.MATRIX_BT601)),
        # ... and on ODD widths (round 4: any width stays on the streaming kernels; the ragged lane is clipped by the buffer resources,
        # the row's last sample goes out as a short, the box filter replicates the last column)
        ("ycc-d32-p3-b10-444-hot-odd515", dict(width=515, height=5, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ,
                                               peak_nits=80, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                                               chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                               color_primaries=pkg.PRIMARIES_BT2020)),
        ("ycc-d32-p3-b12-444-hot-odd1", dict(width=1, height=3, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_PQ,
                                             peak_nits=80, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                                             chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                             color_primaries=pkg.PRIMARIES_BT2020)),
        ("ycc-d32-p3-b12-444-hot-odd1029-clip", dict(width=1029, height=4, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_CLIP,
                                                     alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                                                     matrix_coefficients=pkg.MATRIX_BT601)),
        ("ycc-d32-p3-b10-420-hot-odd1003-avg", dict(width=1003, height=7, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ,
                                                    peak_nits=1000, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                                                    chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                                    color_primaries=pkg.PRIMARIES_BT2020)),
        ("ycc-d32-p3-b12-422-hot-odd777-near", dict(width=777, height=3, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_PQ,
                                                    peak_nits=80, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                                    matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020,
                                                    chroma_downsampling=pkg.DOWNSAMPLE_NEAREST)),
        ("ycc-d32-p3-b10-420-hot-odd513-avg-clip", dict(width=513, height=6, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_CLIP,
                                                        alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                                                        matrix_coefficients=pkg.MATRIX_BT601)),
        ("ycc-d32-p3-b10-422-hot-odd6-avg", dict(width=6, height=2, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_CLIP,
                                                 alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                                 matrix_coefficients=pkg.MATRIX_BT601)),
        # the plug-in's own default save (InitGlobals, AvifFormat.cpp:89,95: 4:2:2, 12 bit; HDR documents: PQ at 80 nits), both
        # output modes, with and without alpha, on a span-multiple width (streaming kernels) and on a ragged one; libheif 1.14's
        # chroma rule (nearest); SDR 16-bit documents carry BT.601 with BT.709 primaries (WriteMetadata.cpp:138-140)
        ("default-d32-p3-b12-422-pq80-near", dict(width=1024, height=6, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_PQ,
                                                  peak_nits=80, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                                  matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020,
                                                  chroma_downsampling=pkg.DOWNSAMPLE_NEAREST)),
        ("default-d32-p3-b12-422-pq80-near-tail", dict(width=1004, height=5, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_PQ,
                                                       peak_nits=80, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                                       matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020,
                                                       chroma_downsampling=pkg.DOWNSAMPLE_NEAREST)),
        ("default-d32-p3-b12-ref-pq80", dict(width=1024, height=6, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_PQ,
                                             peak_nits=80, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)),
        ("default-d32-p4-b12-422-pq80-near", dict(width=1024, height=6, depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_PQ,
                                                  peak_nits=80, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                                  matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020,
                                                  chroma_downsampling=pkg.DOWNSAMPLE_NEAREST)),
        ("default-d32-p4-b12-ref-pq80-premul", dict(width=1024, height=6, depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_PQ,
                                                    peak_nits=80, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_REFERENCE)),
        ("default-d16-p3-b12-422-601-near", dict(width=1024, height=6, depth=16, planes=3, bit_depth=12, alpha_state=pkg.ALPHA_NONE,
                                                 output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422, matrix_coefficients=pkg.MATRIX_BT601,
                                                 color_primaries=pkg.PRIMARIES_BT709, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST)),
        ("default-d8-p3-b12-422-601-near", dict(width=1021, height=5, depth=8, planes=3, bit_depth=12, alpha_state=pkg.ALPHA_NONE,
                                                output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422, matrix_coefficients=pkg.MATRIX_BT601,
                                                color_primaries=pkg.PRIMARIES_BT709, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST)),
        # BASELINE.json config 1 at its real size: 512x512 RGBA8 -> 8-bit 4:2:0 BT.709
        ("baseline-c1-512", dict(width=512, height=512, depth=8, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_STRAIGHT,
                                 output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT709)),
        ("tiny-1x1", dict(width=1, height=1, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ,
                          alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                          matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)),
        ("tiny-3x2-ref", dict(width=3, height=2, depth=16, planes=4, bit_depth=10, alpha_state=pkg.ALPHA_PREMULTIPLIED,
                              output=pkg.OUT_REFERENCE)),
    ]
    # the same branches through the ALIGNED kernel instantiations (every pointer / stride a multiple of 16)
    aligned = []
    for cid, kw in (out + extra)[::3]:
        if kw["width"] == W_ODD:
            k2 = dict(kw, width=128, height=10)
            aligned.append((cid + "-al", k2))
    return out + extra + aligned


def is_float_tier_write(kw):
    """True when the case goes through powf/logf in the reference (tolerance tier T2)."""
    return kw["depth"] == 32 and kw.get("transfer", pkg.TRANSFER_CLIP) != pkg.TRANSFER_CLIP


def read_cases():
    """[(id, kwargs)] -- every ReadHeifImage* / Decode*Row* branch (reference YuvDecode.cpp:55-696,
    ReadHeifImage.cpp:83-1178)."""
    out = []
    alphas = (pkg.ALPHA_NONE, pkg.ALPHA_STRAIGHT, pkg.ALPHA_PREMULTIPLIED)
    mats = [(pkg.MATRIX_BT601, pkg.PRIMARIES_BT709), (pkg.MATRIX_BT709, pkg.PRIMARIES_BT709),
            (pkg.MATRIX_BT2020_NCL, pkg.PRIMARIES_BT2020)]
    # ---- YCbCr -> 8 / 16 bit host (bit-exact tier) ----
    for bits, depth in ((8, 8), (10, 16), (12, 16), (16, 16)):
        for chroma in (pkg.CHROMA_444, pkg.CHROMA_422, pkg.CHROMA_420):
            for a in alphas:
                for (m, pr) in mats:
                    for fr in (1, 0):
                        if fr == 0 and m != pkg.MATRIX_BT709:
                            continue
                        out.append((f"ycc-b{bits}-d{depth}-c{chroma}-a{a}-m{m}-fr{fr}",
                                    dict(width=W_ODD, height=H_ODD, colorspace=pkg.COLORSPACE_YCBCR, chroma=chroma,
                                         bit_depth=bits, depth=depth, alpha_state=a, matrix_coefficients=m,
                                         color_primaries=pr, full_range_flag=fr)))
    out.append(("ycc-b8-d8-nonclx", dict(width=W_ODD, height=H_ODD, colorspace=pkg.COLORSPACE_YCBCR,
                                          chroma=pkg.CHROMA_420, bit_depth=8, depth=8, alpha_state=pkg.ALPHA_NONE,
                                          has_nclx=0, matrix_coefficients=pkg.MATRIX_BT709, full_range_flag=0)))
    out.append(("ycc-b8-d8-gbr-quirk", dict(width=W_ODD, height=H_ODD, colorspace=pkg.COLORSPACE_YCBCR,
                                             chroma=pkg.CHROMA_444, bit_depth=8, depth=8, alpha_state=pkg.ALPHA_NONE,
                                             matrix_coefficients=pkg.MATRIX_RGB_GBR)))
    out.append(("ycc-b10-d16-chromaderived", dict(width=W_ODD, height=H_ODD, colorspace=pkg.COLORSPACE_YCBCR,
                                                   chroma=pkg.CHROMA_422, bit_depth=10, depth=16,
                                                   alpha_state=pkg.ALPHA_NONE,
                                                   matrix_coefficients=pkg.MATRIX_CHROMA_DERIVED_NCL,
                                                   color_primaries=pkg.PRIMARIES_BT2020)))
    out.append(("ycc-b10-d16-ycgco-fallback", dict(width=W_ODD, height=H_ODD, colorspace=pkg.COLORSPACE_YCBCR,
                                                    chroma=pkg.CHROMA_444, bit_depth=10, depth=16,
                                                    alpha_state=pkg.ALPHA_NONE, matrix_coefficients=pkg.MATRIX_YCGCO)))
    # ---- mono ----
    for bits, depth in ((8, 8), (10, 16), (12, 16)):
        for a in alphas:
            for fr in (1, 0):
                out.append((f"mono-b{bits}-d{depth}-a{a}-fr{fr}",
                            dict(width=W_ODD, height=H_ODD, colorspace=pkg.COLORSPACE_MONOCHROME,
                                 chroma=pkg.CHROMA_MONOCHROME, bit_depth=bits, depth=depth, alpha_state=a,
                                 full_range_flag=fr)))
    # ---- planar RGB ----
    for bits, depth in ((8, 8), (10, 16), (12, 16)):
        for a in alphas:
            out.append((f"rgb-b{bits}-d{depth}-a{a}",
                        dict(width=W_ODD, height=H_ODD, colorspace=pkg.COLORSPACE_RGB, chroma=pkg.CHROMA_444,
                             bit_depth=bits, depth=depth, alpha_state=a, matrix_coefficients=pkg.MATRIX_RGB_GBR)))
    # ---- 32-bit host (float tier) ----
    for bits in (10, 12):
        for a in alphas:
            for tc in (pkg.TC_PQ, pkg.TC_HLG, pkg.TC_SMPTE428):
                for cs, chroma in ((pkg.COLORSPACE_YCBCR, pkg.CHROMA_420), (pkg.COLORSPACE_YCBCR, pkg.CHROMA_444),
                                   (pkg.COLORSPACE_RGB, pkg.CHROMA_444)):
                    out.append((f"f32-cs{cs}-b{bits}-c{chroma}-a{a}-tc{tc}",
                                dict(width=W_ODD, height=H_ODD, colorspace=cs, chroma=chroma, bit_depth=bits, depth=32,
                                     alpha_state=a, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                     color_primaries=pkg.PRIMARIES_BT2020, transfer_characteristics=tc,
                                     pq_peak_nits=80, hlg_apply_ootf=0)))
            out.append((f"f32-mono-b{bits}-a{a}-pq",
                        dict(width=W_ODD, height=H_ODD, colorspace=pkg.COLORSPACE_MONOCHROME,
                             chroma=pkg.CHROMA_MONOCHROME, bit_depth=bits, depth=32, alpha_state=a,
                             transfer_characteristics=pkg.TC_PQ, pq_peak_nits=1000)))
    out.append(("f32-ycc-hlg-ootf", dict(width=W_ODD, height=H_ODD, colorspace=pkg.COLORSPACE_YCBCR,
                                          chroma=pkg.CHROMA_422, bit_depth=10, depth=32, alpha_state=pkg.ALPHA_STRAIGHT,
                                          matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020,
                                          transfer_characteristics=pkg.TC_HLG, hlg_apply_ootf=1, hlg_display_gamma=1.2,
                                          hlg_peak_nits=1000)))
    out.append(("f32-rgb-hlg-ootf709", dict(width=W_ODD, height=H_ODD, colorspace=pkg.COLORSPACE_RGB,
                                             chroma=pkg.CHROMA_444, bit_depth=12, depth=32, alpha_state=pkg.ALPHA_NONE,
                                             matrix_coefficients=pkg.MATRIX_RGB_GBR, color_primaries=pkg.PRIMARIES_BT709,
                                             transfer_characteristics=pkg.TC_HLG, hlg_apply_ootf=1,
                                             hlg_display_gamma=1.5, hlg_peak_nits=400)))
    # displayGamma 1.0 is the reference's minimum (AvifFormat.h:49): powf(luma, 0) = 1 also for black pixels
    out.append(("f32-ycc-hlg-ootf-gamma1", dict(width=W_ODD, height=H_ODD, colorspace=pkg.COLORSPACE_YCBCR,
                                                 chroma=pkg.CHROMA_444, bit_depth=10, depth=32, alpha_state=pkg.ALPHA_NONE,
                                                 matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020,
                                                 transfer_characteristics=pkg.TC_HLG, hlg_apply_ootf=1, hlg_display_gamma=1.0,
                                                 hlg_peak_nits=1000)))
    # what the plug-in's default saves decode to: 12-bit 4:2:2, PQ -> RGB f32 (HDR) / BT.601 -> RGB16 (SDR), with and without alpha
    out.append(("default-read-b12-422-pq80-f32", dict(width=1024, height=6, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_422,
                                                       bit_depth=12, depth=32, alpha_state=pkg.ALPHA_NONE,
                                                       matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020,
                                                       transfer_characteristics=pkg.TC_PQ, pq_peak_nits=80)))
    out.append(("default-read-b12-422-pq80-f32-alpha-tail", dict(width=1003, height=5, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_422,
                                                                  bit_depth=12, depth=32, alpha_state=pkg.ALPHA_STRAIGHT,
                                                                  matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020,
                                                                  transfer_characteristics=pkg.TC_PQ, pq_peak_nits=80)))
    out.append(("default-read-b12-422-601-rgb16", dict(width=1024, height=6, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_422,
                                                        bit_depth=12, depth=16, alpha_state=pkg.ALPHA_NONE,
                                                        matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)))
    out.append(("tiny-read-1x1", dict(width=1, height=1, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_420,
                                       bit_depth=8, depth=8, alpha_state=pkg.ALPHA_NONE,
                                       matrix_coefficients=pkg.MATRIX_BT709)))
    aligned = [(cid + "-al", dict(kw, width=128, height=10)) for cid, kw in out[::3] if kw["width"] == W_ODD]
    # largest dynamic-LDS footprint of read_px: 12-bit tables (48 KiB) + RGBA f32 4:2:0 transpose strips (32 KiB)
    aligned.append(("f32-maxlds-b12-420-rgba", dict(width=1024 + 32, height=6, colorspace=pkg.COLORSPACE_YCBCR,
                                                     chroma=pkg.CHROMA_420, bit_depth=12, depth=32,
                                                     alpha_state=pkg.ALPHA_PREMULTIPLIED,
                                                     matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                                                     color_primaries=pkg.PRIMARIES_BT2020,
                                                     transfer_characteristics=pkg.TC_PQ, pq_peak_nits=1000)))
    aligned.append(("ycc-b8-wide-420", dict(width=2048 + 48, height=4, colorspace=pkg.COLORSPACE_YCBCR,
                                             chroma=pkg.CHROMA_420, bit_depth=8, depth=8, alpha_state=pkg.ALPHA_NONE,
                                             matrix_coefficients=pkg.MATRIX_BT709)))
    return out + aligned


def is_float_tier_read(kw):
    return kw["depth"] == 32
