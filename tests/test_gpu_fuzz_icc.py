"""Geometry fuzz BEHIND A DOCUMENT PROFILE: what tests/test_gpu_fuzz_wide.py does for plain saves, with the ICC row
transform in front -- seeded random widths of a few hundred to a few thousand pixels (ragged last spans, one ragged lane), small random
heights, rows padded to 16 bytes, a random even-row tile split, RGB and RGBA documents of 8, 16 and 32 bits, every output kind -- against
the REAL Little CMS 2 driven like ColorProfileConversion.cpp:159-187 (oracle/icc_oracle.c: one cmsDoTransformLineStride per row, in place)
followed by the oracle's pixel loop.  The fixed-width ICC tests (test_gpu_icc.py, test_icc8.py, test_icc16.py) prove the arithmetic on a
handful of widths; the streaming ICC kernels (icc = 1 / 2 / 4) and the table-driven stages of the generic kernel (icc = 3 / 5 / 6 / 7) meet
arbitrary geometries here.  8- and 16-bit documents: bit-exact.  32-bit: |dcode| <= 1 and the small-sample T2 bar of the wide fuzz."""
import ctypes
import functools
import os

import numpy as np
import pytest

import harness
from test_gpu_u8_fast_path import align

pkg = harness.pkg
pytestmark = pytest.mark.gpu
FUZZ_N = int(os.environ.get("AVIFGPU_FUZZ_ICC_N", "120"))
ICC_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle_icc.so")


@functools.lru_cache(maxsize=1)
def _lcms():
    L = ctypes.CDLL(ICC_LIB)
    L.oracle_icc_make_profile.restype = ctypes.c_int32
    L.oracle_icc_make_profile.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_void_p, ctypes.c_uint32]
    L.oracle_icc_make_a2b_profile.restype = ctypes.c_int32
    L.oracle_icc_make_a2b_profile.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32]
    rows = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    for fn in (L.oracle_icc_convert_rows_to_rec2020, L.oracle_icc_convert_rows_to_srgb_float, L.oracle_icc_convert_rows_to_srgb8):
        fn.restype, fn.argtypes = ctypes.c_int32, rows
    L.oracle_icc_convert_rows_to_srgb16.restype = ctypes.c_int32
    L.oracle_icc_convert_rows_to_srgb16.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                                     ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    for name in ("oracle_icc_transform16_open", "oracle_icc_transform8_open"):
        getattr(L, name).restype = ctypes.c_void_p
        getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    L.oracle_icc_transform16_close.argtypes = [ctypes.c_void_p]
    return L


@pytest.fixture(scope="module")
def lcms():
    if not os.path.exists(ICC_LIB):
        pytest.skip("oracle/liboracle_icc.so not built (lcms2 absent)")
    return _lcms()


@functools.lru_cache(maxsize=None)
def _profile(kind, trc, g):
    L = _lcms()
    buf = ctypes.create_string_buffer(1 << 16)
    n = L.oracle_icc_make_profile(kind, trc, g, buf, len(buf))
    assert n > 0
    return buf.raw[:n]


@functools.lru_cache(maxsize=None)
def _a2b_profile(variant):
    L = _lcms()
    buf = ctypes.create_string_buffer(1 << 18)
    n = L.oracle_icc_make_a2b_profile(variant, buf, len(buf))
    assert n > 0
    return buf.raw[:n]


@functools.lru_cache(maxsize=None)
def _table_from_lcms(variant, bits):
    """The table of an A2B profile as the adapter's bridge obtains it (integration/LcmsTableBridge.cpp): from lcms2's own transforms, proven."""
    L, icc = _lcms(), _a2b_profile(variant)
    t = pkg.IccClut16()
    if bits == 8:
        h = L.oracle_icc_transform8_open(icc, len(icc), 0)
        rc = pkg.load().avifgpu_icc_clut8_from_transforms(ctypes.cast(L.oracle_icc_transform16_run_float, ctypes.c_void_p),
                                                          ctypes.cast(L.oracle_icc_transform8_run, ctypes.c_void_p), h, ctypes.byref(t))
    else:
        h = L.oracle_icc_transform16_open(icc, len(icc), 0)
        rc = pkg.load().avifgpu_icc_clut16_from_transforms(ctypes.cast(L.oracle_icc_transform16_run_float, ctypes.c_void_p),
                                                           ctypes.cast(L.oracle_icc_transform16_run, ctypes.c_void_p), h, ctypes.byref(t))
    L.oracle_icc_transform16_close(h)
    assert rc == 0, pkg.load().avifgpu_last_error()
    return t


def _case(i):
    rng = np.random.default_rng(515151 + i)
    depth = int(rng.choice([8, 16, 32, 32]))
    planes = int(rng.choice([3, 3, 4]))
    alpha = pkg.ALPHA_NONE if planes == 3 else int(rng.choice([pkg.ALPHA_STRAIGHT, pkg.ALPHA_PREMULTIPLIED]))
    w = int(rng.integers(40, 3000))
    w = max(8, w - w % int(rng.choice([1, 4, 8, 16])))
    h = int(rng.integers(1, 10))
    kw = dict(width=w, height=h, depth=depth, planes=planes, alpha_state=alpha, chroma=int(rng.choice([pkg.CHROMA_444, pkg.CHROMA_422, pkg.CHROMA_420])),
              output=pkg.OUT_REFERENCE if rng.random() < 0.3 else pkg.OUT_YCBCR, color_primaries=pkg.PRIMARIES_BT709,
              chroma_downsampling=int(rng.choice([pkg.DOWNSAMPLE_AVERAGE, pkg.DOWNSAMPLE_NEAREST])))
    if depth == 32:
        how = str(rng.choice(["linear", "linear", "gamma", "sampled"]))
        sdr = how != "sampled" and rng.random() < 0.3                    # a 32-bit document saved as SDR: -> sRGB, Clip (ColorProfileConversion.cpp:118-123)
        kw.update(bit_depth=int(rng.choice([10, 12])), transfer=pkg.TRANSFER_CLIP if sdr else pkg.TRANSFER_PQ, peak_nits=int(rng.choice([80, 1000])),
                  matrix_coefficients=pkg.MATRIX_BT601 if sdr else pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT709 if sdr else pkg.PRIMARIES_BT2020)
        if alpha == pkg.ALPHA_PREMULTIPLIED and not sdr:
            kw["alpha_state"] = pkg.ALPHA_STRAIGHT                       # premultiply is disabled for HDR saves (Write.cpp:251-257)
    else:
        how = str(rng.choice(["matrix", "matrix", "a2b"]))
        kw.update(bit_depth=int(rng.choice([8, 10, 12])), matrix_coefficients=int(rng.choice([pkg.MATRIX_BT601, pkg.MATRIX_BT709])))
    return kw, how, 2 * int(rng.integers(0, h // 2 + 1)), int(rng.integers(0, 2))


def _gpu_write(gpu, desc, src, row0, nrows, xf):
    import torch
    dev = f"cuda:{gpu.device}"
    H, rowb = src.shape[0], src.shape[1] * src.itemsize
    stride = align(rowb, 16)
    padded = np.full((H, stride), 0x5A, dtype=np.uint8)
    padded[:, :rowb] = src.view(np.uint8).reshape(H, rowb)
    d_src = torch.from_numpy(padded.reshape(-1)).to(dev)
    bufs = harness._alloc_write_out(desc, nrows)
    d_out = {pl: torch.from_numpy(b.view(np.uint8).reshape(-1).copy()).to(dev) for pl, b in bufs.items()}
    ptrs = [d_out[i].data_ptr() if i in d_out else None for i in range(4)]
    strides = [bufs[i].strides[0] if i in bufs else 0 for i in range(4)]
    gpu.write_rows(desc, row0, nrows, d_src.data_ptr() + row0 * stride, stride, ptrs, strides, mem=pkg.MEM_DEVICE,
                   stream=torch.cuda.current_stream(dev).cuda_stream, icc=xf)
    torch.cuda.synchronize(dev)
    raw = {pl: d_out[pl].cpu().numpy().view(bufs[pl].dtype).reshape(bufs[pl].shape) for pl in bufs}
    return harness._trim(desc, raw, nrows, harness.write_planes)


@pytest.mark.parametrize("i", range(FUZZ_N))
def test_write_fuzz_behind_a_profile(gpu, lcms, i):
    kw, how, cut, variant = _case(i)
    d = pkg.WriteDesc(**kw)
    src = harness.make_write_source(d, seed=i)
    has_alpha = int(d.planes == 4)
    conv = src.copy()
    if d.depth == 32:
        sdr = d.transfer == pkg.TRANSFER_CLIP
        kind, trc, g = {"linear": (1, 0, 1.0), "gamma": (3, 0, 2.19921875), "sampled": (1, 3, 1024)}[how]
        icc = _profile(kind, trc, g)
        if how == "gamma":
            src = np.abs(src)                                            # parametric non-linear curves: stay where every lcms2 build agrees
            conv = src.copy()
        target = pkg.ICC_TARGET_SRGB_FLOAT if sdr else pkg.ICC_TARGET_REC2020_LINEAR
        xf = gpu.icc_prepare_sampled(icc, target) if how == "sampled" else gpu.icc_prepare(icc, target)
        fn = lcms.oracle_icc_convert_rows_to_srgb_float if sdr else lcms.oracle_icc_convert_rows_to_rec2020
        assert fn(icc, len(icc), has_alpha, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
    elif d.depth == 16:
        icc = _profile(3, 0, 2.19921875) if how == "matrix" else _a2b_profile(variant)
        xf = gpu.icc_prepare_clut16(icc) if how == "matrix" else _table_from_lcms(variant, 16)
        assert lcms.oracle_icc_convert_rows_to_srgb16(icc, len(icc), has_alpha, 0, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
    else:
        icc = _profile(3, 0, 2.19921875) if how == "matrix" else _a2b_profile(variant)
        xf = gpu.icc_prepare_shaper8(icc) if how == "matrix" else _table_from_lcms(variant, 8)
        assert lcms.oracle_icc_convert_rows_to_srgb8(icc, len(icc), has_alpha, conv.ctypes.data, d.width, d.height, conv.strides[0]) == 0
    for row0, nrows in ((0, cut), (cut, d.height - cut)):
        if nrows == 0:
            continue
        want = harness.oracle_write(d, conv, row0, nrows)
        got = _gpu_write(gpu, d, src, row0, nrows, xf)
        st = harness.compare_write(d, want, got)
        assert "icc=" in gpu.last_kernel(), gpu.last_kernel()
        if d.depth == 32:
            assert st["max_abs"] <= 1, (kw, how, row0, nrows, gpu.last_kernel(), st)
            assert st["exact_frac"] >= 0.98 or st["n"] < 20000, (kw, how, gpu.last_kernel(), st)
        else:
            assert st["max_abs"] == 0, (kw, how, row0, nrows, gpu.last_kernel(), st)
