"""8-bit SDR save path, ICC stage: lcms2's 8-bit matrix-shaper pipeline (document profile -> sRGB, the
transform ColorProfileConversion installs when keepColorProfile is false, ColorProfileConversion.cpp:134-157,:268-331)
reproduced BIT FOR BIT.  Checker: the real Little CMS 2 driven like the reference (oracle/icc_oracle.c).

CPU part: the host-built tables (avifgpu_icc_prepare_shaper8) pushed through a numpy restatement of MatShaperEval16 on
ALL 2^24 RGB triples.  GPU part: the same 2^24 triples through the fused kernel."""
import ctypes
import os

import numpy as np
import pytest

import harness

pkg = harness.pkg
ICC_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle_icc.so")
PROFILES = [("adobergb-g2.2", 3, 0, 2.19921875), ("p3-srgb-trc", 1, 1, 0.0), ("prophoto-d50-g1.8", 2, 0, 1.8),
            ("p3-para-g1.8", 1, 2, 1.8), ("srgb-primaries-linear", 0, 0, 1.0),
            # sampled `curv` tables (lcms2 interpolates them in 16-bit fixed point), incl. a different curve per channel
            ("p3-sampled-srgb-1024", 1, 3, 1024), ("adobergb-sampled-per-channel-256", 3, 4, 256),
            ("prophoto-sampled-per-channel-33", 2, 4, 33),
            # sRGB primaries + power law: the 1.14 matrix degenerates to identity, lcms2 still agrees bit for bit
            ("srgb-primaries-g2.2", 0, 0, 2.2)]


@pytest.fixture(scope="module")
def lcms():
    if not os.path.exists(ICC_LIB):
        pytest.skip("oracle/liboracle_icc.so not built (lcms2 absent)")
    L = ctypes.CDLL(ICC_LIB)
    L.oracle_icc_make_profile.restype = ctypes.c_int32
    L.oracle_icc_make_profile.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_void_p, ctypes.c_uint32]
    L.oracle_icc_convert_rows_to_srgb8.restype = ctypes.c_int32
    L.oracle_icc_convert_rows_to_srgb8.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p,
                                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    L.oracle_icc_make_a2b_profile.restype = ctypes.c_int32
    L.oracle_icc_make_a2b_profile.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32]
    L.oracle_icc_transform8_open.restype = ctypes.c_void_p
    L.oracle_icc_transform8_open.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    L.oracle_icc_transform16_close.argtypes = [ctypes.c_void_p]
    return L


def _profile(L, kind, trc, g):
    buf = ctypes.create_string_buffer(1 << 18)
    n = L.oracle_icc_make_profile(kind, trc, g, buf, len(buf))
    assert n > 0
    return buf.raw[:n]


def _all_rgb():
    r, g, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    return np.stack([r, g, b], axis=-1).reshape(4096, 4096 * 3).copy()


def _shaper(icc):
    t = pkg.IccShaper8()
    rc = pkg.load().avifgpu_icc_prepare_shaper8(icc, len(icc), ctypes.byref(t))
    assert rc == 0, pkg.load().avifgpu_last_error()
    return t


@pytest.mark.parametrize("name,kind,trc,g", [PROFILES[0], PROFILES[1], PROFILES[6]])
def test_tables_reproduce_lcms2_on_every_rgb_triple(lcms, name, kind, trc, g):
    icc = _profile(lcms, kind, trc, g)
    sh = _shaper(icc)
    src = _all_rgb()
    want = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb8(icc, len(icc), 0, want.ctypes.data, 4096, 4096, want.strides[0]) == 0
    s1 = np.array(sh.shaper1, dtype=np.int32)
    M = np.array(sh.matrix, dtype=np.int32)
    s2 = np.array(sh.shaper2, dtype=np.uint8)
    px_all, want_all = src.reshape(-1, 3), want.reshape(-1, 3)
    step = 1 << 20                                           # chunked: 16 M-element int64 temporaries cost more in page faults than in math
    for lo in range(0, len(px_all), step):
        px = px_all[lo:lo + step]
        R, G, B = s1[0][px[:, 0]], s1[1][px[:, 1]], s1[2][px[:, 2]]
        for i in range(3):                                   # MatShaperEval16 (products fit int32: |m| * 16384 * 3 < 2^31 for these profiles)
            l = np.clip((M[i, 0] * R + M[i, 1] * G + M[i, 2] * B + 0x2000) >> 14, 0, 16384)
            assert np.array_equal(s2[i][l], want_all[lo:lo + step, i]), (name, lo, i)


def test_prepare_rejects_non_profiles():
    t = pkg.IccShaper8()
    assert pkg.load().avifgpu_icc_prepare_shaper8(bytes(300), 300, ctypes.byref(t)) == pkg.formatCannotRead


def _gpu(gpu, d, src, sh, pad=False):
    import torch
    dev = f"cuda:{gpu.device}"
    bufs = harness._alloc_write_out(d, d.height)
    if pad:                                        # rows padded to a 16-byte multiple (+16): aligned rows whatever the width
        stride = (src.shape[1] + 15) // 16 * 16 + 16
        wide = np.full((src.shape[0], stride), 0xA5, dtype=src.dtype)
        wide[:, :src.shape[1]] = src
        src = wide[:, :src.shape[1]]
        d_src = torch.from_numpy(wide.reshape(-1)).to(dev)
    else:
        d_src = torch.from_numpy(src.reshape(-1)).to(dev)
    d_out = {pl: torch.from_numpy(b.view(np.uint8).reshape(-1).copy()).to(dev) for pl, b in bufs.items()}
    ptrs = [d_out[i].data_ptr() if i in d_out else None for i in range(4)]
    strides = [bufs[i].strides[0] if i in bufs else 0 for i in range(4)]
    gpu.write_rows(d, 0, d.height, d_src.data_ptr(), src.strides[0], ptrs, strides, mem=pkg.MEM_DEVICE,
                   stream=torch.cuda.current_stream(dev).cuda_stream, icc=sh)
    torch.cuda.synchronize(dev)
    for pl in bufs:
        bufs[pl] = d_out[pl].cpu().numpy().view(bufs[pl].dtype).reshape(bufs[pl].shape)
    return harness._trim(d, bufs, d.height, harness.write_planes)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,trc,g", PROFILES)
def test_gpu_every_rgb_triple_bit_exact(gpu, lcms, name, kind, trc, g):
    """All 2^24 RGB8 triples: fused ICC + 8-bit copy (the reference hand-off) == lcms2 ConvertRow then the pixel loop."""
    icc = _profile(lcms, kind, trc, g)
    sh = gpu.icc_prepare_shaper8(icc)
    src = _all_rgb()
    want = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb8(icc, len(icc), 0, want.ctypes.data, 4096, 4096, want.strides[0]) == 0
    d = pkg.WriteDesc(width=4096, height=4096, depth=8, planes=3, bit_depth=8, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
    got = _gpu(gpu, d, src, sh)
    assert np.array_equal(got[0], want), name
    assert "icc=3" in gpu.last_kernel()


@pytest.mark.gpu
def test_gpu_icc8_then_every_output_kind(gpu, lcms):
    """RGBA (alpha copied), premultiply, 10-bit rescale and fused YCbCr 4:2:0 after the ICC stage, ragged size, bit-exact."""
    icc = _profile(lcms, 3, 0, 2.19921875)
    sh = gpu.icc_prepare_shaper8(icc)
    for kw in (dict(planes=4, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_REFERENCE),
               dict(planes=4, bit_depth=10, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                    matrix_coefficients=pkg.MATRIX_BT601),
               dict(planes=3, bit_depth=8, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                    matrix_coefficients=pkg.MATRIX_BT601)):
        d = pkg.WriteDesc(width=333, height=41, depth=8, **kw)
        src = harness.make_write_source(d, seed=21)
        conv = src.copy()
        assert lcms.oracle_icc_convert_rows_to_srgb8(icc, len(icc), int(d.planes == 4), conv.ctypes.data, d.width, d.height,
                                                     conv.strides[0]) == 0
        want = harness.oracle_write(d, conv)
        got = _gpu(gpu, d, src, sh)
        for pl in want:
            assert np.array_equal(got[pl], want[pl]), (kw, pl)


@pytest.mark.gpu
@pytest.mark.parametrize("width", [1024, 1000, 336, 20])
@pytest.mark.parametrize("chroma", ["444", "422", "420"])
@pytest.mark.parametrize("planes,alpha_state", [(3, "NONE"), (4, "STRAIGHT"), (4, "PREMULTIPLIED")])
def test_gpu_icc8_packed_u8_plane_path(gpu, lcms, width, chroma, planes, alpha_state):
    """The default kind of save (8-bit document -> u8 Y, Cb, Cr(, A) planes) with the ICC stage in front runs on the packed
    16-pixel footprint kernel (aligned rows): every chroma format, both down-sampling modes, alpha copied / premultiplied
    after the transform, widths with and without a ragged last footprint, odd heights -- bit-exact against lcms2 + the oracle."""
    icc = _profile(lcms, 3, 0, 2.19921875)
    sh = gpu.icc_prepare_shaper8(icc)
    for ds in (pkg.DOWNSAMPLE_AVERAGE, pkg.DOWNSAMPLE_NEAREST):
        d = pkg.WriteDesc(width=width, height=37, depth=8, planes=planes, bit_depth=8, alpha_state=getattr(pkg, "ALPHA_" + alpha_state),
                          output=pkg.OUT_YCBCR, chroma=getattr(pkg, "CHROMA_" + chroma), chroma_downsampling=ds,
                          matrix_coefficients=pkg.MATRIX_BT601)
        src = harness.make_write_source(d, seed=width + planes)
        conv = src.copy()
        assert lcms.oracle_icc_convert_rows_to_srgb8(icc, len(icc), int(planes == 4), conv.ctypes.data, d.width, d.height,
                                                     conv.strides[0]) == 0
        want = harness.oracle_write(d, conv)
        got = _gpu(gpu, d, src, sh, pad=True)
        name = gpu.last_kernel()
        assert "icc=3" in name and "aligned=1" in name, name
        for pl in want:
            assert np.array_equal(got[pl], want[pl]), (ds, pl, name)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,trc,g", [PROFILES[0], PROFILES[2], PROFILES[-1]])
def test_gpu_every_rgb_triple_through_the_packed_path_both_matrix_forms(gpu, lcms, name, kind, trc, g):
    """All 2^24 RGB8 triples through the packed u8-plane kernel (8-bit YCbCr 4:4:4 behind the ICC stage), once with two products of a
    matrix row in a v_dot2_i32_i16 (the default where the operands fit 16 bits) and once as three 24-bit mads (tuning bit 32):
    the same planes, equal to lcms2 ConvertRow + the oracle's pixel loop."""
    icc = _profile(lcms, kind, trc, g)
    sh = gpu.icc_prepare_shaper8(icc)
    src = _all_rgb()
    conv = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb8(icc, len(icc), 0, conv.ctypes.data, 4096, 4096, conv.strides[0]) == 0
    d = pkg.WriteDesc(width=4096, height=4096, depth=8, planes=3, bit_depth=8, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                      chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT601)
    want = harness.oracle_write(d, conv)
    try:
        for variant in (1 | 2 | 4, (1 | 2 | 4) | 32):
            gpu.lib.avifgpu_set_hot_variant(variant)
            got = _gpu(gpu, d, src, sh, pad=True)
            assert "icc=3" in gpu.last_kernel() and "aligned=1" in gpu.last_kernel()
            for pl in want:
                assert np.array_equal(got[pl], want[pl]), (name, variant, pl)
    finally:
        gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4)


@pytest.mark.gpu
def test_host_shim_converts_8bit_document_to_srgb(gpu, lcms):
    """FormatRecord shim with saveOptions.convertToSRGB: tiles converted with lcms2's 8-bit pipeline and handed off like
    CreateHeifImageRGBEightBit (interleaved RGBA, heif_chroma_interleaved_RGBA), byte-identical to lcms2 + the pixel loop."""
    from fake_host import FakeHost
    H = pkg.host
    icc = _profile(lcms, 3, 0, 2.19921875)
    d = pkg.WriteDesc(width=517, height=67, depth=8, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_REFERENCE)
    src = harness.make_write_source(d, seed=5)
    conv = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb8(icc, len(icc), 1, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
    want = harness.oracle_write(d, conv)
    host = FakeHost(d.width, d.height, 8, 4, max_data=517 * 4 * 2 * 10, image=src)
    keep = ctypes.create_string_buffer(icc, len(icc))
    host.fr.iCCprofileData = ctypes.cast(keep, ctypes.c_void_p)
    host.fr.iCCprofileSize = len(icc)
    opts = H.SaveUIOptions(imageBitDepth=8, hdrTransferFunction=pkg.TRANSFER_CLIP, pq=H.PQOptions(1000),
                           chromaSubsampling=pkg.CHROMA_420, lossless=0, convertToRec2020=0, convertToSRGB=1)
    img = H.Image()
    code = gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host.fr), pkg.ALPHA_STRAIGHT, ctypes.byref(opts), pkg.OUT_REFERENCE,
                                                  pkg.MATRIX_BT601, pkg.PRIMARIES_BT709, ctypes.byref(img))
    assert code == 0, gpu.lib.avifgpu_last_error()
    assert len(host.rects) > 3                                     # really tiled
    raw = (ctypes.c_uint8 * (img.stride[0] * d.height)).from_address(img.plane[0])
    got = np.frombuffer(raw, dtype=np.uint8).reshape(d.height, img.stride[0])[:, :d.width * 4]
    assert np.array_equal(got, want[0])
    gpu.lib.avifgpu_image_free(ctypes.byref(img))


# ---- round 6: LUT-based (A2B) document profiles at 8 bit -- the 33^3 table of the caller's own transforms, evaluated like PrelinEval8 ----
def _a2b_profile(L, variant):
    buf = ctypes.create_string_buffer(1 << 18)
    n = L.oracle_icc_make_a2b_profile(variant, buf, len(buf))
    assert n > 0
    return buf.raw[:n]


def _clut8_from_transform(L, icc, extra_flags=0):
    """What the adapter does (integration/LcmsTableBridge.cpp): cmsDoTransform on the TYPE_RGB_8 transform it owns and on a float twin, as callbacks."""
    h = L.oracle_icc_transform8_open(icc, len(icc), extra_flags)
    assert h
    t = pkg.IccClut16()
    rc = pkg.load().avifgpu_icc_clut8_from_transforms(ctypes.cast(L.oracle_icc_transform16_run_float, ctypes.c_void_p),
                                                      ctypes.cast(L.oracle_icc_transform8_run, ctypes.c_void_p), h, ctypes.byref(t))
    L.oracle_icc_transform16_close(h)
    return rc, t


@pytest.mark.parametrize("variant", [0, 1])
def test_a2b_table_from_the_callers_8bit_transform_reproduces_lcms2_on_every_rgb_triple(lcms, variant):
    """CPU: the table read out through the callbacks + a numpy restatement of PrelinEval8 (word 257 b, TetrahedralInterp16, FROM_16_TO_8)
    against the real library's TYPE_RGB_8 transform on ALL 2^24 triples."""
    import test_icc16
    icc = _a2b_profile(lcms, variant)
    rc, t = _clut8_from_transform(lcms, icc)
    assert rc == 0, pkg.load().avifgpu_last_error()
    table = np.ctypeslib.as_array(t.table).reshape(33, 33, 33, 4)[..., :3].copy()
    src = _all_rgb()
    want = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb8(icc, len(icc), 0, want.ctypes.data, 4096, 4096, want.strides[0]) == 0
    px_all, want_all = src.reshape(-1, 3), want.reshape(-1, 3)
    step = 1 << 20
    for lo in range(0, len(px_all), step):
        w = test_icc16._tetrahedral(table, px_all[lo:lo + step].astype(np.int64) * 257)
        got = ((w * 65281 + 8388608) >> 24).astype(np.uint8)
        assert np.array_equal(got, want_all[lo:lo + step]), (variant, lo)


def test_8bit_table_read_out_refuses_what_is_not_that_table(lcms):
    """A matrix/TRC profile runs lcms2's matrix-shaper at 8 bit (different arithmetic: avifgpu_icc_prepare_shaper8 covers it) and
    cmsFLAGS_NOOPTIMIZE evaluates the profile's own pipeline: the proof on 16384 probe colours must refuse both, with the reason."""
    icc = _profile(lcms, 3, 0, 2.19921875)                     # AdobeRGB-like matrix/TRC
    rc, _ = _clut8_from_transform(lcms, icc)
    assert rc == pkg.formatCannotRead and b"avifgpu_icc_prepare_shaper8" in pkg.load().avifgpu_last_error()
    rc, _ = _clut8_from_transform(lcms, _a2b_profile(lcms, 0), extra_flags=0x0100)
    assert rc == pkg.formatCannotRead


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1])
def test_gpu_a2b_profile_every_rgb_triple_bit_exact(gpu, lcms, variant):
    """All 2^24 RGB triples of an 8-bit document behind a LUT-based profile through write_px<..., icc = 7> (interleaved hand-off = the converted
    row itself) == the real lcms2's TYPE_RGB_8 transform."""
    import torch
    icc = _a2b_profile(lcms, variant)
    rc, t = _clut8_from_transform(lcms, icc)
    assert rc == 0
    src = _all_rgb()
    want = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb8(icc, len(icc), 0, want.ctypes.data, 4096, 4096, want.strides[0]) == 0
    d = pkg.WriteDesc(width=4096, height=4096, depth=8, planes=3, bit_depth=8, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
    dev = f"cuda:{gpu.device}"
    d_src = torch.from_numpy(src).to(dev)
    d_out = torch.zeros_like(d_src)
    gpu.write_rows(d, 0, 4096, d_src.data_ptr(), d_src.stride(0), [d_out.data_ptr(), None, None, None], [d_out.stride(0), 0, 0, 0],
                   mem=pkg.MEM_DEVICE, stream=torch.cuda.current_stream(dev).cuda_stream, icc=t)
    torch.cuda.synchronize(dev)
    assert "icc=7" in gpu.last_kernel(), gpu.last_kernel()
    assert np.array_equal(d_out.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(planes=3, bit_depth=8, output=1, chroma=2, chroma_downsampling=1, matrix_coefficients=6),
                                dict(planes=4, bit_depth=8, alpha_state=2, output=1, chroma=1, matrix_coefficients=6),
                                dict(planes=4, bit_depth=12, alpha_state=1, output=1, chroma=3, matrix_coefficients=6),
                                dict(planes=3, bit_depth=10, output=0)])
def test_gpu_a2b_profile_then_every_output_kind(gpu, lcms, kw):
    """The table stage in front of the rest of the 8-bit pixel loop (rescale to 10 / 12 bit, premultiply, stage B, alpha copied): the oracle's
    pixel loop on rows the real lcms2 converted == the fused kernel; odd width and height, host pointers through the tile scheduler."""
    icc = _a2b_profile(lcms, 1)
    rc, t = _clut8_from_transform(lcms, icc)
    assert rc == 0
    d = pkg.WriteDesc(width=1001, height=37, depth=8, **kw)
    src = harness.make_write_source(d, seed=3)
    conv = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb8(icc, len(icc), 1 if d.planes == 4 else 0, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
    want = harness.oracle_write(d, conv)
    bufs = harness._alloc_write_out(d, d.height)
    ptrs = [bufs[i].ctypes.data if i in bufs else None for i in range(4)]
    strides = [bufs[i].strides[0] if i in bufs else 0 for i in range(4)]
    gpu.write_rows(d, 0, d.height, src.ctypes.data, src.strides[0], ptrs, strides, mem=pkg.MEM_HOST, icc=t)
    got = harness._trim(d, bufs, d.height, harness.write_planes)
    assert "icc=7" in gpu.last_kernel(), gpu.last_kernel()
    for pl in want:
        assert np.array_equal(got[pl], want[pl]), (kw, pl)


BRIDGE = os.path.join(os.path.dirname(ICC_LIB), "..", "avif-format_amd", "libavifgpu_lcms_bridge.so")


def _bridge_table8(icc):
    """integration/LcmsTableBridge.cpp::avifgpu_lcms_document_to_srgb_clut8 -- the glue the plug-in's adapter compiles."""
    if not os.path.exists(BRIDGE):
        pytest.skip("libavifgpu_lcms_bridge.so not built (lcms2 absent)")
    pkg.load()
    B = ctypes.CDLL(BRIDGE)
    B.avifgpu_lcms_document_to_srgb_clut8.restype = ctypes.c_int32
    B.avifgpu_lcms_document_to_srgb_clut8.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(pkg.IccClut16)]
    t = pkg.IccClut16()
    rc = B.avifgpu_lcms_document_to_srgb_clut8(icc, len(icc) if icc is not None else 0, ctypes.byref(t) if icc is not None else None)
    return rc, t


def test_adapter_bridge_builds_the_8bit_table(lcms):
    for variant in (0, 1):
        icc = _a2b_profile(lcms, variant)
        rc, t = _bridge_table8(icc)
        assert rc == 0, pkg.load().avifgpu_last_error()
        assert np.array_equal(np.ctypeslib.as_array(t.table), np.ctypeslib.as_array(_clut8_from_transform(lcms, icc)[1].table))
    assert _bridge_table8(_profile(lcms, 3, 0, 2.19921875))[0] == pkg.formatCannotRead     # matrix/TRC: lcms2 runs its matrix-shaper, the proof refuses
    assert _bridge_table8(bytes(400))[0] == pkg.formatCannotRead                          # not a profile
    assert _bridge_table8(None)[0] == pkg.formatBadParameters


@pytest.mark.gpu
@pytest.mark.parametrize("planes", [3, 4])
def test_host_shim_converts_a_lut_based_8bit_document_with_the_callers_table(gpu, lcms, planes):
    """The adapter's flow for an A2B profile at 8 bit (round 6): the plain entry refuses the profile (formatCannotRead from the matrix/TRC
    parser), the bridge computes the table from lcms2's own transforms, avifgpu_host_create_heif_image_with_table converts with it --
    the decision still made like the plug-in's (keepColorProfile: no conversion, the table is ignored)."""
    from fake_host import FakeHost
    H = pkg.host
    icc = _a2b_profile(lcms, 1)
    alpha = pkg.ALPHA_STRAIGHT if planes == 4 else pkg.ALPHA_NONE
    d = pkg.WriteDesc(width=389, height=53, depth=8, planes=planes, bit_depth=8, alpha_state=alpha, output=pkg.OUT_REFERENCE)
    src = harness.make_write_source(d, seed=9)
    conv = src.copy()
    assert lcms.oracle_icc_convert_rows_to_srgb8(icc, len(icc), int(planes == 4), conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
    want = harness.oracle_write(d, conv)
    keep = ctypes.create_string_buffer(icc, len(icc))
    opts = H.SaveUIOptions(imageBitDepth=8, hdrTransferFunction=pkg.TRANSFER_CLIP, pq=H.PQOptions(1000), chromaSubsampling=pkg.CHROMA_420,
                           lossless=0, keepColorProfile=0, iccDecision=H.ICC_LIKE_PLUGIN)

    def save(table):
        host = FakeHost(d.width, d.height, 8, planes, max_data=d.width * planes * 9, image=src)
        host.fr.iCCprofileData = ctypes.cast(keep, ctypes.c_void_p)
        host.fr.iCCprofileSize = len(icc)
        img = H.Image()
        code = gpu.lib.avifgpu_host_create_heif_image_with_table(ctypes.byref(host.fr), alpha, ctypes.byref(opts), pkg.OUT_REFERENCE,
                                                                 pkg.MATRIX_BT601, pkg.PRIMARIES_BT709,
                                                                 ctypes.byref(table) if table is not None else None, ctypes.byref(img))
        return code, img

    def plane0(img):
        raw = (ctypes.c_uint8 * (img.stride[0] * d.height)).from_address(img.plane[0])
        return np.frombuffer(raw, dtype=np.uint8).reshape(d.height, img.stride[0])[:, :d.width * planes].copy()

    code, img = save(None)
    assert code == pkg.formatCannotRead
    gpu.lib.avifgpu_image_free(ctypes.byref(img))
    rc, table = _bridge_table8(icc)
    assert rc == 0, gpu.lib.avifgpu_last_error()
    code, img = save(table)
    assert code == 0, gpu.lib.avifgpu_last_error()
    assert "icc=7" in gpu.last_kernel(), gpu.last_kernel()
    assert np.array_equal(plane0(img), want[0])
    gpu.lib.avifgpu_image_free(ctypes.byref(img))
    opts.keepColorProfile = 1                                             # the decision says "no conversion": the table is ignored
    code, img = save(table)
    assert code == 0
    assert np.array_equal(plane0(img), harness.oracle_write(d, src)[0])
    gpu.lib.avifgpu_image_free(ctypes.byref(img))
