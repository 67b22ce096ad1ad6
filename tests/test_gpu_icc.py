"""the ICC row transform of the HDR save path on the GPU vs the REAL Little CMS 2 (the third-party
library the reference calls, present in this image), driven exactly as the reference drives it
(oracle/icc_oracle.c = ColorProfileConversion.cpp:98-132,159-187,235-266 + ColorProfileGeneration.cpp:141-178).

Tolerance: the transformed linear values feed the PQ curve and a truncating quantiser, so the bar is the write tier's:
|delta code| <= 1 and >= 99 % exact; the transform itself is also checked in isolation (Clip curve at 12 bit is a
x4095 magnifier of the float result)."""
import ctypes
import os

import numpy as np
import pytest

import harness

pkg = harness.pkg
pytestmark = pytest.mark.gpu

ICC_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle_icc.so")


@pytest.fixture(scope="module")
def lcms():
    if not os.path.exists(ICC_LIB):
        pytest.skip("oracle/liboracle_icc.so not built (lcms2 absent)")
    L = ctypes.CDLL(ICC_LIB)
    L.oracle_icc_make_profile.restype = ctypes.c_int32
    L.oracle_icc_make_profile.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_void_p, ctypes.c_uint32]
    L.oracle_icc_convert_rows_to_rec2020.restype = ctypes.c_int32
    L.oracle_icc_convert_rows_to_rec2020.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p,
                                                      ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    L.oracle_icc_convert_rows_to_srgb_float.restype = ctypes.c_int32
    L.oracle_icc_convert_rows_to_srgb_float.argtypes = L.oracle_icc_convert_rows_to_rec2020.argtypes
    return L


def _profile(L, kind, trc, g):
    buf = ctypes.create_string_buffer(1 << 16)
    n = L.oracle_icc_make_profile(kind, trc, g, buf, len(buf))
    assert n > 0
    return buf.raw[:n]


PROFILES = [("srgb-linear", 0, 0, 1.0), ("p3-linear", 1, 0, 1.0), ("prophoto-linear-d50", 2, 0, 1.0),
            ("adobergb-gamma2.2", 3, 0, 2.19921875), ("srgb-parametric", 0, 1, 0.0), ("p3-para-gamma1.8", 1, 2, 1.8)]


@pytest.mark.parametrize("name,kind,trc,g", PROFILES)
@pytest.mark.parametrize("planes", [3, 4])
def test_icc_then_pq_matches_lcms2(gpu, lcms, name, kind, trc, g, planes):
    icc = _profile(lcms, kind, trc, g)
    xf = gpu.icc_prepare(icc)
    alpha = pkg.ALPHA_STRAIGHT if planes == 4 else pkg.ALPHA_NONE
    for output, chroma, bits, transfer in ((pkg.OUT_YCBCR, pkg.CHROMA_444, 10, pkg.TRANSFER_PQ),
                                           (pkg.OUT_YCBCR, pkg.CHROMA_420, 12, pkg.TRANSFER_PQ),
                                           (pkg.OUT_REFERENCE, pkg.CHROMA_444, 12, pkg.TRANSFER_CLIP)):
        d = pkg.WriteDesc(width=515, height=18, depth=32, planes=planes, bit_depth=bits, transfer=transfer, peak_nits=80,
                          alpha_state=alpha, output=output, chroma=chroma, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                          color_primaries=pkg.PRIMARIES_BT2020)
        src = harness.make_write_source(d, seed=5)
        if trc != 0 or g != 1.0:
            src = np.abs(src)                     # non-linear curves: stay where every lcms2 build agrees (no negative inputs)
        # reference flow: lcms2 converts each row in place, then the pixel loop runs on the converted row
        conv = src.copy()
        assert lcms.oracle_icc_convert_rows_to_rec2020(icc, len(icc), int(planes == 4), conv.ctypes.data, d.width, d.height,
                                                       conv.strides[0]) == 0
        want = harness.oracle_write(d, conv)
        # GPU: one fused launch on the ORIGINAL rows
        got = _gpu_write_icc(gpu, d, src, xf)
        st = harness.compare_write(d, want, got)
        print(f"icc->rec2020 {name} planes {planes} out {output} {bits}-bit transfer {transfer}: exact {st['exact_frac']:.5f} max {st['max_abs']}")
        assert st["max_abs"] <= 1, (name, output, st)
        assert harness.t2_exact_ok(st, harness.T2_MIN_EXACT_ICC), (name, output, st)
        assert ("icc=1" if (trc == 0 and g == 1.0) else "icc=2") in gpu.last_kernel()


@pytest.mark.parametrize("name,kind,trc,g", [p for p in PROFILES if p[2] == 0 and p[3] == 1.0])
@pytest.mark.parametrize("width", [1024, 516, 8])
def test_icc1_streaming_kernel_matches_lcms2(gpu, lcms, name, kind, trc, g, width):
    """The usual HDR save: a 32-bit document with a LINEAR profile (sRGB / Display P3 / ProPhoto primaries) -> Rec.2020 -> PQ ->
    4:4:4 planes takes the streaming kernel with the matrix in front (fp32 FMAs; lcms2: double accumulation).  Same bars."""
    icc = _profile(lcms, kind, trc, g)
    xf = gpu.icc_prepare(icc)
    for bits, transfer, peak, chroma in ((10, pkg.TRANSFER_PQ, 80, pkg.CHROMA_444), (12, pkg.TRANSFER_PQ, 1000, pkg.CHROMA_444),
                                         (12, pkg.TRANSFER_CLIP, 80, pkg.CHROMA_444), (10, pkg.TRANSFER_HLG, 80, pkg.CHROMA_444),
                                         (10, pkg.TRANSFER_PQ, 80, pkg.CHROMA_420), (12, pkg.TRANSFER_PQ, 1000, pkg.CHROMA_422),
                                         (12, pkg.TRANSFER_CLIP, 80, pkg.CHROMA_420)):
        d = pkg.WriteDesc(width=width, height=9, depth=32, planes=3, bit_depth=bits, transfer=transfer, peak_nits=peak,
                          alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=chroma,
                          matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
        src = harness.make_write_source(d, seed=width + bits)
        conv = src.copy()
        assert lcms.oracle_icc_convert_rows_to_rec2020(icc, len(icc), 0, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
        want = harness.oracle_write(d, conv)
        got = _gpu_write_icc(gpu, d, src, xf)
        k = gpu.last_kernel()
        assert ("write_rgb32_icc1_ycbcr444_hot" in k) if chroma == pkg.CHROMA_444 else ("write_rgb32_ycbcr_sub_hot" in k and "icc=1" in k), k
        st = harness.compare_write(d, want, got)
        print(f"icc1-streaming {name} width {width} {bits}-bit transfer {transfer} chroma {chroma}: exact {st['exact_frac']:.5f} max {st['max_abs']}")
        assert st["max_abs"] <= 1, (name, st)
        assert harness.t2_exact_ok(st, harness.T2_MIN_EXACT_ICC), (name, st)


@pytest.mark.parametrize("name,kind,trc,g", PROFILES)
@pytest.mark.parametrize("width", [1024, 520, 516, 8])
def test_icc_streaming_reference_handoff_matches_lcms2(gpu, lcms, name, kind, trc, g, width):
    """Round 5: the interleaved RRGGBB hand-off (AVIFGPU_OUT_REFERENCE -- what integration/ asks for by default) behind a linear or a
    simple parametric document profile runs on the streaming ICC kernel (write_rgb32_icc1_ycbcr444_hot<..., OUTREF>: the codes leave
    through the wave's strip as coalesced stores) where the plane rows are 16-byte aligned; same arithmetic as the 4:4:4 form, same bars."""
    icc = _profile(lcms, kind, trc, g)
    xf = gpu.icc_prepare(icc)
    for bits, transfer, peak in ((10, pkg.TRANSFER_PQ, 80), (12, pkg.TRANSFER_PQ, 1000), (12, pkg.TRANSFER_CLIP, 80), (10, pkg.TRANSFER_HLG, 80),
                                 (12, pkg.TRANSFER_SMPTE428, 80)):
        d = pkg.WriteDesc(width=width, height=11, depth=32, planes=3, bit_depth=bits, transfer=transfer, peak_nits=peak,
                          alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE, chroma=pkg.CHROMA_444,
                          matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
        src = harness.make_write_source(d, seed=width + bits)
        if trc != 0 or g != 1.0:
            src = np.abs(src)
        conv = src.copy()
        assert lcms.oracle_icc_convert_rows_to_rec2020(icc, len(icc), 0, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
        want = harness.oracle_write(d, conv)
        got = _gpu_write_icc(gpu, d, src, xf)
        k = gpu.last_kernel()
        if (width * 6) % 16 == 0:
            assert "write_rgb32_icc1_ycbcr444_hot" in k and "out=ref" in k, k
        st = harness.compare_write(d, want, got)
        print(f"icc-ref-streaming {name} width {width} {bits}-bit transfer {transfer}: exact {st['exact_frac']:.5f} max {st['max_abs']}  {k}")
        assert st["max_abs"] <= 1, (name, k, st)
        assert harness.t2_exact_ok(st, harness.T2_MIN_EXACT_ICC), (name, k, st)
        # ... and the generic kernel (tuning word 0: no streaming kernels) on the same rows agrees within the tier
        gpu.lib.avifgpu_set_hot_variant(0)
        try:
            ref = _gpu_write_icc(gpu, d, src, xf)
            assert "write_px" in gpu.last_kernel(), gpu.last_kernel()
        finally:
            gpu.lib.avifgpu_set_hot_variant(7)
        st2 = harness.compare_write(d, ref, got)
        assert st2["max_abs"] <= 1 and harness.t2_exact_ok(st2, 0.99), (name, st2)


@pytest.mark.parametrize("name,kind,trc,g", [p for p in PROFILES if not (p[2] == 0 and p[3] == 1.0)])
@pytest.mark.parametrize("width", [1024, 516])
def test_icc2_streaming_kernels_match_lcms2(gpu, lcms, name, kind, trc, g, width):
    """A 32-bit document whose profile has ONE parametric / gamma curve for R, G and B (gamma 2.2, the sRGB curve, gamma 1.8 as `para`)
    saved as Rec.2100 PQ: the curve runs on the streaming kernels, per sample as loaded, as exp2(g log2(a R + b)) -- the same
    arithmetic the generic icc = 2 kernel uses (AG_ICC_FASTPOW), so the same bars against the real lcms2."""
    icc = _profile(lcms, kind, trc, g)
    xf = gpu.icc_prepare(icc)
    for planes, bits, transfer, peak, chroma in ((3, 10, pkg.TRANSFER_PQ, 80, pkg.CHROMA_444), (3, 12, pkg.TRANSFER_PQ, 1000, pkg.CHROMA_444),
                                                 (3, 10, pkg.TRANSFER_PQ, 80, pkg.CHROMA_420), (3, 12, pkg.TRANSFER_SMPTE428, 80, pkg.CHROMA_422),
                                                 (4, 12, pkg.TRANSFER_PQ, 80, pkg.CHROMA_444)):
        d = pkg.WriteDesc(width=width, height=10, depth=32, planes=planes, bit_depth=bits, transfer=transfer, peak_nits=peak,
                          alpha_state=pkg.ALPHA_STRAIGHT if planes == 4 else pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=chroma,
                          matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
        src = np.abs(harness.make_write_source(d, seed=width + bits))      # non-linear curves: no negative inputs (see above)
        conv = src.copy()
        assert lcms.oracle_icc_convert_rows_to_rec2020(icc, len(icc), int(planes == 4), conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
        want = harness.oracle_write(d, conv)
        got = _gpu_write_icc(gpu, d, src, xf)
        k = gpu.last_kernel()
        assert "hot" in k and "icc=2" in k, k
        st = harness.compare_write(d, want, got)
        print(f"icc2-streaming {name} width {width} planes {planes} {bits}-bit transfer {transfer} chroma {chroma}: exact {st['exact_frac']:.5f} max {st['max_abs']}")
        assert st["max_abs"] <= 1, (name, st)
        assert harness.t2_exact_ok(st, harness.T2_MIN_EXACT_ICC), (name, st)
        # ... and within tier 2 of the generic kernel (FP64-free there too: same curve arithmetic, matrix in fp32 on both)
        try:
            gpu.lib.avifgpu_set_hot_variant(0)
            slow = _gpu_write_icc(gpu, d, src, xf)
            assert "write_px" in gpu.last_kernel() and "icc=2" in gpu.last_kernel(), gpu.last_kernel()
        finally:
            gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4)
        st2 = harness.compare_write(d, slow, got)
        assert st2["max_abs"] <= 1 and st2["exact_frac"] >= 0.995, (name, st2)


@pytest.mark.parametrize("name,kind,trc,g", [p for p in PROFILES if p[2] == 0 and p[3] == 1.0])
@pytest.mark.parametrize("width,alpha", [(1024, pkg.ALPHA_STRAIGHT), (515, pkg.ALPHA_PREMULTIPLIED), (7, pkg.ALPHA_STRAIGHT)])
def test_icc1_rgba_streaming_kernel_matches_lcms2(gpu, lcms, name, kind, trc, g, width, alpha):
    """RGBA f32 with a linear profile -> Rec.2020 PQ -> Y,Cb,Cr,A: the C5 kernel with the matrix on R,G,B in front (alpha copied)."""
    icc = _profile(lcms, kind, trc, g)
    xf = gpu.icc_prepare(icc)
    for bits, transfer in ((12, pkg.TRANSFER_PQ), (10, pkg.TRANSFER_CLIP)):
        d = pkg.WriteDesc(width=width, height=9, depth=32, planes=4, bit_depth=bits, transfer=transfer, peak_nits=80,
                          alpha_state=alpha, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                          matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
        src = harness.make_write_source(d, seed=width + bits)
        conv = src.copy()
        assert lcms.oracle_icc_convert_rows_to_rec2020(icc, len(icc), 1, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
        want = harness.oracle_write(d, conv)
        got = _gpu_write_icc(gpu, d, src, xf)
        k = gpu.last_kernel()
        assert "write_rgba32_ycbcra444_hot" in k and "icc=1" in k, k
        st = harness.compare_write(d, want, got)
        print(f"icc1-rgba-streaming {name} width {width} {bits}-bit transfer {transfer}: exact {st['exact_frac']:.5f} max {st['max_abs']}")
        assert st["max_abs"] <= 1, (name, st)
        assert harness.t2_exact_ok(st, harness.T2_MIN_EXACT_ICC), (name, st)


@pytest.mark.parametrize("name,kind,trc,g", [p for p in PROFILES if p[2] == 0 and p[3] == 1.0])
@pytest.mark.parametrize("width,alpha", [(1024, pkg.ALPHA_PREMULTIPLIED), (515, pkg.ALPHA_STRAIGHT)])
def test_icc4_rgba_streaming_kernel_matches_lcms2(gpu, lcms, name, kind, trc, g, width, alpha):
    """RGBA f32 with a linear profile saved as SDR: -> sRGB (matrix + inverse curve) -> Clip -> Y,Cb,Cr,A on the RGBA streaming kernel."""
    icc = _profile(lcms, kind, trc, g)
    xf = gpu.icc_prepare(icc, pkg.ICC_TARGET_SRGB_FLOAT)
    for bits in (10, 12):
        d = pkg.WriteDesc(width=width, height=9, depth=32, planes=4, bit_depth=bits, transfer=pkg.TRANSFER_CLIP,
                          alpha_state=alpha, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT601,
                          color_primaries=pkg.PRIMARIES_BT709)
        src = harness.make_write_source(d, seed=width + bits)
        conv = src.copy()
        assert lcms.oracle_icc_convert_rows_to_srgb_float(icc, len(icc), 1, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
        want = harness.oracle_write(d, conv)
        got = _gpu_write_icc(gpu, d, src, xf)
        k = gpu.last_kernel()
        assert "write_rgba32_ycbcra444_hot" in k and "icc=4" in k, k
        st = harness.compare_write(d, want, got)
        print(f"icc4-rgba-streaming {name} width {width} {bits}-bit: exact {st['exact_frac']:.5f} max {st['max_abs']}")
        assert st["max_abs"] <= 1 and harness.t2_exact_ok(st, harness.T2_MIN_EXACT_ICC), (name, st)


@pytest.mark.parametrize("name,kind,trc,g", [p for p in PROFILES if p[2] == 0 and p[3] == 1.0])
@pytest.mark.parametrize("width", [1024, 516])
def test_icc4_streaming_kernel_matches_lcms2(gpu, lcms, name, kind, trc, g, width):
    """The SDR save of a 32-bit document with a linear profile (always converted to sRGB, ColorProfileConversion.cpp:118-123):
    matrix + inverse sRGB curve in front of the Clip quantiser, on the streaming kernels (single-precision curve)."""
    icc = _profile(lcms, kind, trc, g)
    xf = gpu.icc_prepare(icc, pkg.ICC_TARGET_SRGB_FLOAT)
    worst = 1.0
    for bits, chroma in ((12, pkg.CHROMA_444), (10, pkg.CHROMA_420), (12, pkg.CHROMA_422)):
        d = pkg.WriteDesc(width=width, height=9, depth=32, planes=3, bit_depth=bits, transfer=pkg.TRANSFER_CLIP,
                          alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=chroma, matrix_coefficients=pkg.MATRIX_BT601,
                          color_primaries=pkg.PRIMARIES_BT709)
        src = harness.make_write_source(d, seed=width + bits)
        conv = src.copy()
        assert lcms.oracle_icc_convert_rows_to_srgb_float(icc, len(icc), 0, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
        want = harness.oracle_write(d, conv)
        got = _gpu_write_icc(gpu, d, src, xf)
        k = gpu.last_kernel()
        assert "icc=4" in k and ("write_rgb32_icc1_ycbcr444_hot" in k if chroma == pkg.CHROMA_444 else "write_rgb32_ycbcr_sub_hot" in k), k
        st = harness.compare_write(d, want, got)
        print(f"icc4-streaming {name} width {width} {bits}-bit chroma {chroma}: exact {st['exact_frac']:.5f} max {st['max_abs']}")
        assert st["max_abs"] <= 1, (name, st)
        worst = min(worst, st["exact_frac"])
    assert worst >= harness.T2_MIN_EXACT_ICC, (name, worst)


@pytest.mark.parametrize("name,kind,trc,g", PROFILES)
@pytest.mark.parametrize("planes", [3, 4])
def test_icc_to_srgb_then_clip_matches_lcms2(gpu, lcms, name, kind, trc, g, planes):
    """The SDR save of a 32-bit document: always converted to sRGB with TYPE_RGB[A]_FLT (ColorProfileConversion.cpp:118-123,
    :268-331), then the Clip branch of the pixel loop.  [TRC] -> 3x3 -> inverse sRGB curve, all in FP64 like lcms2."""
    icc = _profile(lcms, kind, trc, g)
    xf = gpu.icc_prepare(icc, pkg.ICC_TARGET_SRGB_FLOAT)
    assert xf.out_curve == 4
    alpha = pkg.ALPHA_PREMULTIPLIED if planes == 4 else pkg.ALPHA_NONE
    worst = 1.0
    for output, chroma, bits in ((pkg.OUT_REFERENCE, pkg.CHROMA_444, 10), (pkg.OUT_REFERENCE, pkg.CHROMA_444, 12),
                                 (pkg.OUT_YCBCR, pkg.CHROMA_420, 12)):
        d = pkg.WriteDesc(width=515, height=18, depth=32, planes=planes, bit_depth=bits, transfer=pkg.TRANSFER_CLIP,
                          alpha_state=alpha, output=output, chroma=chroma, matrix_coefficients=pkg.MATRIX_BT601,
                          color_primaries=pkg.PRIMARIES_BT709)
        src = harness.make_write_source(d, seed=9)
        if trc != 0 or g != 1.0:
            src = np.abs(src)
        conv = src.copy()
        assert lcms.oracle_icc_convert_rows_to_srgb_float(icc, len(icc), int(planes == 4), conv.ctypes.data, d.width, d.height,
                                                          conv.strides[0]) == 0
        want = harness.oracle_write(d, conv)
        got = _gpu_write_icc(gpu, d, src, xf)
        st = harness.compare_write(d, want, got)
        print(f"icc->srgb {name} planes {planes} out {output} {bits}-bit: exact {st['exact_frac']:.5f} max {st['max_abs']}")
        assert st["max_abs"] <= 1, (name, output, st)
        worst = min(worst, st["exact_frac"])
        assert "icc=4" in gpu.last_kernel()
    assert worst >= harness.T2_MIN_EXACT_ICC, (name, worst)
    # the sRGB target belongs to the Clip save only (the reference never builds it for PQ / SMPTE 428)
    d = pkg.WriteDesc(width=16, height=2, depth=32, planes=planes, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=1000,
                      alpha_state=alpha, output=pkg.OUT_REFERENCE)
    with pytest.raises(pkg.AvifGpuError) as e:
        _gpu_write_icc(gpu, d, harness.make_write_source(d), xf)
    assert e.value.code == pkg.formatBadParameters


def _gpu_write_icc(gpu, d, src, xf):
    import torch
    dev = f"cuda:{gpu.device}"
    bufs = harness._alloc_write_out(d, d.height)
    d_src = torch.from_numpy(src.view(np.uint8).reshape(-1)).to(dev)
    d_out = {pl: torch.from_numpy(b.view(np.uint8).reshape(-1).copy()).to(dev) for pl, b in bufs.items()}
    ptrs = [d_out[i].data_ptr() if i in d_out else None for i in range(4)]
    strides = [bufs[i].strides[0] if i in bufs else 0 for i in range(4)]
    gpu.write_rows(d, 0, d.height, d_src.data_ptr(), src.strides[0], ptrs, strides, mem=pkg.MEM_DEVICE,
                   stream=torch.cuda.current_stream(dev).cuda_stream, icc=xf)
    torch.cuda.synchronize(dev)
    for pl in bufs:
        bufs[pl] = d_out[pl].cpu().numpy().view(bufs[pl].dtype).reshape(bufs[pl].shape)
    return harness._trim(d, bufs, d.height, harness.write_planes)


def test_icc_streaming_and_generic_kernels_agree_within_tier2(gpu, lcms):
    """with an ICC transform the streaming kernels (single-precision 3x3) and the generic kernel (FP64 3x3) are
    identical within tier 2, not byte for byte."""
    icc = _profile(lcms, 1, 0, 1.0)                                  # linear Display P3
    for target, transfer, bits in ((None, pkg.TRANSFER_PQ, 12), (pkg.ICC_TARGET_SRGB_FLOAT, pkg.TRANSFER_CLIP, 12)):
        xf = gpu.icc_prepare(icc) if target is None else gpu.icc_prepare(icc, target)
        for planes, chroma in ((3, pkg.CHROMA_444), (3, pkg.CHROMA_420), (4, pkg.CHROMA_444)):
            d = pkg.WriteDesc(width=1024, height=16, depth=32, planes=planes, bit_depth=bits, transfer=transfer, peak_nits=80,
                              alpha_state=pkg.ALPHA_STRAIGHT if planes == 4 else pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=chroma,
                              matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
            src = harness.make_write_source(d, seed=77)
            try:
                fast = _gpu_write_icc(gpu, d, src, xf)
                assert "hot" in gpu.last_kernel(), gpu.last_kernel()
                gpu.lib.avifgpu_set_hot_variant(0)
                slow = _gpu_write_icc(gpu, d, src, xf)
                assert "write_px" in gpu.last_kernel(), gpu.last_kernel()
            finally:
                gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4)
            st = harness.compare_write(d, slow, fast)
            assert st["max_abs"] <= 1 and st["exact_frac"] >= 0.995, (target, planes, chroma, st)


def test_icc_matrix_agrees_with_lcms2(gpu, lcms):
    """The prepared 3x3 reproduces lcms2 on the unit vectors to float rounding, for every test profile."""
    for name, kind, trc, g in PROFILES:
        if trc != 0 or g != 1.0:
            continue
        icc = _profile(lcms, kind, trc, g)
        xf = gpu.icc_prepare(icc)
        rows = np.eye(3, dtype=np.float32).reshape(1, 9).copy()
        assert lcms.oracle_icc_convert_rows_to_rec2020(icc, len(icc), 0, rows.ctypes.data, 3, 1, rows.strides[0]) == 0
        m = np.array(list(xf.matrix)).reshape(3, 3)
        assert np.allclose(rows.reshape(3, 3).T, m, rtol=0, atol=6e-8), name


def test_icc_prepare_rejects_what_it_cannot_do(gpu):
    bad = bytearray(200)
    assert gpu.lib.avifgpu_icc_prepare(bytes(bad), len(bad), 0, ctypes.byref(pkg.IccTransform())) == pkg.formatCannotRead
    d = pkg.WriteDesc(width=8, height=2, depth=8, planes=3, bit_depth=8, output=pkg.OUT_REFERENCE)
    xf = pkg.IccTransform()
    xf.trc_type[0] = xf.trc_type[1] = xf.trc_type[2] = 1
    src = harness.make_write_source(d)
    with pytest.raises(pkg.AvifGpuError) as e:
        _gpu_write_icc(gpu, d, src, xf)
    assert e.value.code == pkg.formatBadParameters


def test_host_shim_converts_document_profile(gpu, lcms):
    """The FormatRecord shim with saveOptions.convertToRec2020: document profile bytes in formatRecord->iCCprofileData,
    tiles converted + encoded in one launch each; result == lcms2 ConvertRow per row followed by the pixel loop."""
    from fake_host import FakeHost
    H = pkg.host
    icc = _profile(lcms, 1, 0, 1.0)                       # Display-P3 primaries, linear: a typical 32-bit document
    d = pkg.WriteDesc(width=300, height=40, depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=1000,
                      alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                      chroma_downsampling=pkg.DOWNSAMPLE_NEAREST,          # the shim's default: libheif 1.14.0's co-sited sample
                      matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
    src = harness.make_write_source(d, seed=11)
    conv = src.copy()
    assert lcms.oracle_icc_convert_rows_to_rec2020(icc, len(icc), 1, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
    want = harness.oracle_write(d, conv)

    host = FakeHost(d.width, d.height, 32, 4, max_data=300 * 16 * 9, image=src)
    keep = ctypes.create_string_buffer(icc, len(icc))
    host.fr.iCCprofileData = ctypes.cast(keep, ctypes.c_void_p)
    host.fr.iCCprofileSize = len(icc)
    opts = H.SaveUIOptions(imageBitDepth=12, hdrTransferFunction=pkg.TRANSFER_PQ, pq=H.PQOptions(1000),
                           chromaSubsampling=pkg.CHROMA_422, lossless=0, convertToRec2020=1)
    img = H.Image()
    code = gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host.fr), pkg.ALPHA_STRAIGHT, ctypes.byref(opts), pkg.OUT_YCBCR,
                                                  pkg.MATRIX_BT2020_NCL, pkg.PRIMARIES_BT2020, ctypes.byref(img))
    assert code == 0, gpu.lib.avifgpu_last_error()
    got = {}
    for pl, (w, xs, ys) in harness.write_planes(d).items():
        h = (d.height + ys) >> ys
        raw = (ctypes.c_uint8 * (img.stride[pl] * h)).from_address(img.plane[pl])
        got[pl] = np.frombuffer(raw, dtype=np.uint8).reshape(h, img.stride[pl])[:, :w * 2].view(np.uint16).copy()
    st = harness.compare_write(d, want, got)
    assert st["max_abs"] <= 1 and harness.t2_exact_ok(st, harness.T2_MIN_EXACT_ICC), st
    gpu.lib.avifgpu_image_free(ctypes.byref(img))
    # a profile the GPU stage cannot take (not an ICC blob) surfaces as an error so the caller can keep lcms2
    host2 = FakeHost(d.width, d.height, 32, 4, image=src)
    junk = ctypes.create_string_buffer(256)
    host2.fr.iCCprofileData = ctypes.cast(junk, ctypes.c_void_p)
    host2.fr.iCCprofileSize = 256
    img2 = H.Image()
    assert gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host2.fr), pkg.ALPHA_STRAIGHT, ctypes.byref(opts), pkg.OUT_YCBCR,
                                                  pkg.MATRIX_BT2020_NCL, pkg.PRIMARIES_BT2020, ctypes.byref(img2)) == pkg.formatCannotRead


def test_host_shim_sdr_save_of_32bit_document(gpu, lcms):
    """FormatRecord shim, 32-bit document + transfer Clip + saveOptions.convertToSRGB: lcms2's float pipeline to sRGB fused
    in front of the Clip branch, tile by tile."""
    from fake_host import FakeHost
    H = pkg.host
    icc = _profile(lcms, 3, 0, 1.0)                       # AdobeRGB primaries, linear (what a 32-bit AdobeRGB document embeds)
    d = pkg.WriteDesc(width=300, height=40, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_CLIP,
                      alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                      chroma_downsampling=pkg.DOWNSAMPLE_NEAREST,
                      matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)
    src = harness.make_write_source(d, seed=13)
    conv = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb_float(icc, len(icc), 0, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
    want = harness.oracle_write(d, conv)
    host = FakeHost(d.width, d.height, 32, 3, max_data=300 * 12 * 10, image=src)
    keep = ctypes.create_string_buffer(icc, len(icc))
    host.fr.iCCprofileData = ctypes.cast(keep, ctypes.c_void_p)
    host.fr.iCCprofileSize = len(icc)
    opts = H.SaveUIOptions(imageBitDepth=12, hdrTransferFunction=pkg.TRANSFER_CLIP, pq=H.PQOptions(1000),
                           chromaSubsampling=pkg.CHROMA_420, lossless=0, convertToRec2020=0, convertToSRGB=1)
    img = H.Image()
    code = gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host.fr), pkg.ALPHA_NONE, ctypes.byref(opts), pkg.OUT_YCBCR,
                                                  pkg.MATRIX_BT601, pkg.PRIMARIES_BT709, ctypes.byref(img))
    assert code == 0, gpu.lib.avifgpu_last_error()
    assert len(host.rects) == 4
    got = {}
    for pl, (w, xs, ys) in harness.write_planes(d).items():
        h = (d.height + ys) >> ys
        raw = (ctypes.c_uint8 * (img.stride[pl] * h)).from_address(img.plane[pl])
        got[pl] = np.frombuffer(raw, dtype=np.uint8).reshape(h, img.stride[pl])[:, :w * 2].view(np.uint16).copy()
    st = harness.compare_write(d, want, got)
    assert st["max_abs"] <= 1 and harness.t2_exact_ok(st, harness.T2_MIN_EXACT_ICC), st
    gpu.lib.avifgpu_image_free(ctypes.byref(img))


@pytest.mark.parametrize("keep", [1, 0])
def test_host_shim_decides_like_the_plugin_clip_and_kept_profile(gpu, lcms, keep):
    """iccDecision = LIKE_PLUGIN: the shim takes ColorProfileConversion's decision itself (ColorProfileConversion.cpp:98-157).
    A 32-bit Clip save with keepColorProfile installs NO transform (:105) -- the planes equal the oracle's conversion of the
    UNTOUCHED document -- and without it the document is converted to sRGB first (:118-123), profile bytes identical."""
    from fake_host import FakeHost
    H = pkg.host
    icc = _profile(lcms, 3, 0, 1.0)                       # AdobeRGB primaries, linear
    d = pkg.WriteDesc(width=300, height=40, depth=32, planes=3, bit_depth=12, transfer=pkg.TRANSFER_CLIP,
                      alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
    src = harness.make_write_source(d, seed=29)
    conv = src.copy()
    if not keep:
        assert lcms.oracle_icc_convert_rows_to_srgb_float(icc, len(icc), 0, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
    want = harness.oracle_write(d, conv)
    host = FakeHost(d.width, d.height, 32, 3, max_data=300 * 12 * 10, image=src)
    blob = ctypes.create_string_buffer(icc, len(icc))
    host.fr.iCCprofileData = ctypes.cast(blob, ctypes.c_void_p)
    host.fr.iCCprofileSize = len(icc)
    # convertToSRGB = 1 is deliberately left set: LIKE_PLUGIN must override the explicit flags
    opts = H.SaveUIOptions(imageBitDepth=12, hdrTransferFunction=pkg.TRANSFER_CLIP, pq=H.PQOptions(1000), chromaSubsampling=pkg.CHROMA_444,
                           convertToSRGB=1, keepColorProfile=keep, iccDecision=H.ICC_LIKE_PLUGIN)
    assert gpu.lib.avifgpu_host_required_conversion_for_record(ctypes.byref(host.fr), ctypes.byref(opts)) == (H.CONVERT_NONE if keep else H.CONVERT_TO_SRGB)
    img = H.Image()
    code = gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host.fr), pkg.ALPHA_NONE, ctypes.byref(opts), pkg.OUT_REFERENCE,
                                                  -1, -1, ctypes.byref(img))
    assert code == 0, gpu.lib.avifgpu_last_error()
    raw = (ctypes.c_uint8 * (img.stride[0] * d.height)).from_address(img.plane[0])
    got = {0: np.frombuffer(raw, dtype=np.uint8).reshape(d.height, img.stride[0])[:, :d.width * 3 * 2].view(np.uint16).copy()}
    st = harness.compare_write(d, want, got)
    if keep:
        assert st["max_abs"] == 0, st                     # no ICC arithmetic at all: the plain Clip path, bit-exact
    else:
        assert st["max_abs"] <= 1 and harness.t2_exact_ok(st, harness.T2_MIN_EXACT_ICC), st
    gpu.lib.avifgpu_image_free(ctypes.byref(img))
    # a profile that cannot be opened where the reference opens it: runtime_error -> writErr + the reference's message
    junk = ctypes.create_string_buffer(b"\0" * 256)
    host2 = FakeHost(d.width, d.height, 32, 3, image=src)
    host2.fr.iCCprofileData = ctypes.cast(junk, ctypes.c_void_p)
    host2.fr.iCCprofileSize = 256
    img2 = H.Image()
    code = gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host2.fr), pkg.ALPHA_NONE, ctypes.byref(opts), pkg.OUT_REFERENCE, -1, -1, ctypes.byref(img2))
    if keep:
        assert code == 0                                   # never opened (:107)
        gpu.lib.avifgpu_image_free(ctypes.byref(img2))
    else:
        assert code == pkg.writErr and b"Unable to load the document color profile." in gpu.lib.avifgpu_last_error()


SAMPLED = [("p3-sampled-srgb-1024", 1, 3, 1024), ("adobergb-sampled-per-channel-256", 3, 4, 256), ("prophoto-sampled-per-channel-33", 2, 4, 33)]
# profiles that MIX the two kinds of curve: (name, kind, trc, entries, parametric_mask)
MIXED = [("p3-R-sampled-G-srgb-para-B-gamma2.2", 1, 5, 1024, 0b110), ("adobergb-R-linear-GB-sampled-256", 3, 6, 256, 0b001),
         ("srgb-R-sampled-4096-G-para-B-gamma", 0, 5, 4096, 0b110)]


@pytest.mark.gpu
def test_sampled_curve_tables_equal_lcms2_per_word(lcms):
    """avifgpu_icc_prepare_sampled (host only): curve[c][w] is what lcms2's float curve stage returns for every 16-bit word w -- checked
    through the real library by converting gray ramps v = w / 65535 with an identity-primaries profile is not possible (the matrix
    follows), so the check is structural: the table equals cmsEvalToneCurve16's output / 65535 as restated in icc_profile.cpp, it is
    monotone for the monotone test curves, and its ends are 0 and 1."""
    lib = pkg.load()
    for name, kind, trc, g in SAMPLED:
        icc = _profile(lcms, kind, trc, g)
        t = pkg.IccSampled32()
        assert lib.avifgpu_icc_prepare_sampled(icc, len(icc), pkg.ICC_TARGET_REC2020_LINEAR, ctypes.byref(t)) == 0, name
        assert lib.avifgpu_icc_prepare(icc, len(icc), pkg.ICC_TARGET_REC2020_LINEAR, ctypes.byref(pkg.IccTransform())) == pkg.formatCannotRead
        c = np.ctypeslib.as_array(t.curve)
        assert c.shape == (3, 65536) and np.all(np.diff(c, axis=1) >= 0), name
        assert np.all(c[:, 0] == 0.0) and np.all(c[:, 65535] == 1.0), name
        # the profile's own tables ride along, and what the kernel does with them in LDS (icc_sampled_curve_lds: LinLerp1D with the
        # division by 0xffff, then w / 65535 as a multiply and an FMA on a two-float reciprocal) returns curve[] bit for bit, every word
        n = list(t.entries)
        assert n == [int(g)] * 3, (name, n)
        w = np.arange(65536, dtype=np.uint64)
        for ch in range(3):
            tab = np.ctypeslib.as_array(t.table16)[ch, :n[ch]].astype(np.uint64)
            pairs = tab | (np.append(tab[1:], tab[-1:]) << np.uint64(16))
            x = np.uint64(n[ch] - 1) * w
            y = x + np.uint64(0x7fff)
            val3 = (y - np.uint64(0x7fff)) + ((y + (y >> np.uint64(16)) + np.uint64(1)) >> np.uint64(16))      # the kernel's form of the division
            assert np.array_equal(val3, x + (x + np.uint64(0x7fff)) // np.uint64(0xffff))
            pr = pairs[(val3 >> np.uint64(16)).astype(np.int64)]
            y0, y1 = pr & np.uint64(0xffff), pr >> np.uint64(16)
            dif = ((y1.astype(np.int64) - y0.astype(np.int64)) * (val3 & np.uint64(0xffff)).astype(np.int64) + 0x8000) & 0xffffffff
            out16 = ((dif >> 16) + y0.astype(np.int64)) & 0xffff
            wf = out16.astype(np.float32)
            rh = np.float32(1.0 / 65535.0)
            rl = np.float32(1.0 / 65535.0 - float(rh))
            fma = lambda a, b, cc: (a.astype(np.float64) * b.astype(np.float64) + cc.astype(np.float64)).astype(np.float32)   # exact product + one rounding
            got = fma(wf, np.full_like(wf, rh), wf * rl)
            assert np.array_equal(got.view(np.uint32), c[ch].view(np.uint32)), (name, ch)
    # an all-parametric profile is not "sampled"; a non-profile is rejected
    icc = _profile(lcms, 1, 2, 1.8)
    assert lib.avifgpu_icc_prepare_sampled(icc, len(icc), 0, ctypes.byref(pkg.IccSampled32())) == pkg.formatCannotRead
    assert lib.avifgpu_icc_prepare_sampled(bytes(200), 200, 0, ctypes.byref(pkg.IccSampled32())) == pkg.formatCannotRead


def test_single_precision_word_equals_the_librarys_for_every_float():
    """tools/satword_check (built from tools/satword_check.hip with hipcc): the kernel's single-precision form of
    _cmsQuickSaturateWord(v * 65535.0) against the double-precision one, all 2^32 float bit patterns, on the device."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "satword_check")
    if not os.path.exists(exe):
        pytest.skip("tools/satword_check not built (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/satword_check tools/satword_check.hip)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "differing words: 0" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("name,kind,trc,g", SAMPLED)
@pytest.mark.parametrize("planes", [3, 4])
def test_sampled_document_curves_match_lcms2(gpu, lcms, name, kind, trc, g, planes):
    """A 32-bit document whose profile carries sampled `curv` tables (one curve for all channels, or a different one per channel):
    lcms2's float pipeline quantises every sample to a 16-bit word and interpolates the table in fixed point; the GPU forms the same
    word and looks the tabulated result up (icc = 6).  Against the REAL lcms2 + the oracle's pixel loop, tier-2 bars."""
    icc = _profile(lcms, kind, trc, g)
    alpha = pkg.ALPHA_STRAIGHT if planes == 4 else pkg.ALPHA_NONE
    for target, conv_fn, cfgs in (
            (pkg.ICC_TARGET_REC2020_LINEAR, lcms.oracle_icc_convert_rows_to_rec2020,
             ((pkg.OUT_YCBCR, pkg.CHROMA_444, 10, pkg.TRANSFER_PQ), (pkg.OUT_YCBCR, pkg.CHROMA_420, 12, pkg.TRANSFER_PQ), (pkg.OUT_REFERENCE, pkg.CHROMA_444, 12, pkg.TRANSFER_SMPTE428))),
            (pkg.ICC_TARGET_SRGB_FLOAT, lcms.oracle_icc_convert_rows_to_srgb_float,
             ((pkg.OUT_YCBCR, pkg.CHROMA_422, 12, pkg.TRANSFER_CLIP), (pkg.OUT_REFERENCE, pkg.CHROMA_444, 10, pkg.TRANSFER_CLIP)))):
        xf = gpu.icc_prepare_sampled(icc, target)
        for output, chroma, bits, transfer in cfgs:
            hdr = transfer != pkg.TRANSFER_CLIP
            d = pkg.WriteDesc(width=516, height=18, depth=32, planes=planes, bit_depth=bits, transfer=transfer, peak_nits=80,
                              alpha_state=alpha, output=output, chroma=chroma,
                              matrix_coefficients=pkg.MATRIX_BT2020_NCL if hdr else pkg.MATRIX_BT601,
                              color_primaries=pkg.PRIMARIES_BT2020 if hdr else pkg.PRIMARIES_BT709)
            src = harness.make_write_source(d, seed=41)       # negatives, > 1 and NaN-free: lcms2 saturates the word at both ends
            conv = src.copy()
            assert conv_fn(icc, len(icc), int(planes == 4), conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
            want = harness.oracle_write(d, conv)
            got = _gpu_write_icc(gpu, d, src, xf)
            assert "icc=6" in gpu.last_kernel() and " lds" in gpu.last_kernel(), gpu.last_kernel()
            try:                                               # the memory-lookup form of the same kernel (tuning bit 64): the same planes
                gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4 | 64)
                mem = _gpu_write_icc(gpu, d, src, xf)
                assert "icc=6" in gpu.last_kernel() and " lds" not in gpu.last_kernel(), gpu.last_kernel()
            finally:
                gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4)
            for pl in got:
                assert np.array_equal(got[pl], mem[pl]), (name, pl)
            st = harness.compare_write(d, want, got)
            print(f"icc-sampled {name} planes {planes} target {target} out {output} {bits}-bit transfer {transfer}: exact {st['exact_frac']:.5f} max {st['max_abs']}")
            assert st["max_abs"] <= 1, (name, st)
            assert harness.t2_exact_ok(st, harness.T2_MIN_EXACT_ICC), (name, st)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,trc,n,mask", MIXED)
@pytest.mark.parametrize("planes", [3, 4])
def test_mixed_sampled_and_parametric_curves_match_lcms2(gpu, lcms, name, kind, trc, n, mask, planes):
    """A 32-bit document whose profile mixes sampled and parametric channels: lcms2's curves stage evaluates each channel by its own
    kind (cmsEvalToneCurveFloat: the 16-bit word for a table, the double formula on the unquantised float for a segment), and so does
    the icc = 6 kernel, channel by channel.  Against the REAL lcms2 + the oracle's pixel loop, tier-2 bars; LDS and memory forms agree."""
    icc = _profile(lcms, kind, trc, n)
    alpha = pkg.ALPHA_STRAIGHT if planes == 4 else pkg.ALPHA_NONE
    for target, conv_fn, cfgs in (
            (pkg.ICC_TARGET_REC2020_LINEAR, lcms.oracle_icc_convert_rows_to_rec2020,
             ((pkg.OUT_YCBCR, pkg.CHROMA_444, 10, pkg.TRANSFER_PQ), (pkg.OUT_YCBCR, pkg.CHROMA_422, 12, pkg.TRANSFER_PQ))),
            (pkg.ICC_TARGET_SRGB_FLOAT, lcms.oracle_icc_convert_rows_to_srgb_float,
             ((pkg.OUT_YCBCR, pkg.CHROMA_420, 12, pkg.TRANSFER_CLIP), (pkg.OUT_REFERENCE, pkg.CHROMA_444, 10, pkg.TRANSFER_CLIP)))):
        xf = gpu.icc_prepare_sampled(icc, target)
        assert xf.parametric_mask == mask
        for output, chroma, bits, transfer in cfgs:
            hdr = transfer != pkg.TRANSFER_CLIP
            d = pkg.WriteDesc(width=517, height=18, depth=32, planes=planes, bit_depth=bits, transfer=transfer, peak_nits=80,
                              alpha_state=alpha, output=output, chroma=chroma, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST if chroma == pkg.CHROMA_422 else 0,
                              matrix_coefficients=pkg.MATRIX_BT2020_NCL if hdr else pkg.MATRIX_BT601,
                              color_primaries=pkg.PRIMARIES_BT2020 if hdr else pkg.PRIMARIES_BT709)
            src = harness.make_write_source(d, seed=47)
            conv = src.copy()
            assert conv_fn(icc, len(icc), int(planes == 4), conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
            want = harness.oracle_write(d, conv)
            got = _gpu_write_icc(gpu, d, src, xf)
            lds = n <= 4096
            assert "icc=6" in gpu.last_kernel() and (" lds" in gpu.last_kernel()) == lds, gpu.last_kernel()
            try:
                gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4 | 64)
                mem = _gpu_write_icc(gpu, d, src, xf)
                assert "icc=6" in gpu.last_kernel() and " lds" not in gpu.last_kernel(), gpu.last_kernel()
            finally:
                gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4)
            for pl in got:
                assert np.array_equal(got[pl], mem[pl]), (name, pl)
            st = harness.compare_write(d, want, got)
            print(f"icc-mixed {name} planes {planes} target {target} out {output} {bits}-bit transfer {transfer}: exact {st['exact_frac']:.5f} max {st['max_abs']}")
            assert st["max_abs"] <= 1, (name, st)
            assert harness.t2_exact_ok(st, harness.T2_MIN_EXACT_ICC), (name, st)


def test_host_shim_takes_sampled_profiles_itself(gpu, lcms):
    """convertToRec2020 with a sampled-curve profile: the shim falls through from the parametric form to the tabulated one instead of
    returning formatCannotRead (which is now left to LUT-based profiles)."""
    from fake_host import FakeHost
    H = pkg.host
    icc = _profile(lcms, 1, 3, 1024)
    d = pkg.WriteDesc(width=300, height=40, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                      alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
    src = np.abs(harness.make_write_source(d, seed=43))
    conv = src.copy()
    assert lcms.oracle_icc_convert_rows_to_rec2020(icc, len(icc), 0, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
    want = harness.oracle_write(d, conv)
    host = FakeHost(d.width, d.height, 32, 3, max_data=300 * 12 * 10, image=src)
    blob = ctypes.create_string_buffer(icc, len(icc))
    host.fr.iCCprofileData = ctypes.cast(blob, ctypes.c_void_p)
    host.fr.iCCprofileSize = len(icc)
    opts = H.SaveUIOptions(imageBitDepth=10, hdrTransferFunction=pkg.TRANSFER_PQ, pq=H.PQOptions(80), chromaSubsampling=pkg.CHROMA_444,
                           iccDecision=H.ICC_LIKE_PLUGIN)
    img = H.Image()
    code = gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host.fr), pkg.ALPHA_NONE, ctypes.byref(opts), pkg.OUT_REFERENCE, -1, -1, ctypes.byref(img))
    assert code == 0, gpu.lib.avifgpu_last_error()
    raw = (ctypes.c_uint8 * (img.stride[0] * d.height)).from_address(img.plane[0])
    got = {0: np.frombuffer(raw, dtype=np.uint8).reshape(d.height, img.stride[0])[:, :d.width * 3 * 2].view(np.uint16).copy()}
    st = harness.compare_write(d, want, got)
    assert st["max_abs"] <= 1 and harness.t2_exact_ok(st, harness.T2_MIN_EXACT_ICC), st
    gpu.lib.avifgpu_image_free(ctypes.byref(img))
