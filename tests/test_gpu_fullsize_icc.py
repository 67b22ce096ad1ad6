"""SURVEY 8(f)-1 at BASELINE size: the ICC row transform fused into the save kernels, WHOLE 8192 x 8192 frames against the REAL
Little CMS 2 (oracle/icc_oracle.c drives it call for call like ColorProfileConversion.cpp:159-187: one cmsDoTransformLineStride per
row, in place) followed by the oracle's pixel loop on every host core (WriteHeifImage.cpp:1031-1135).  The small-frame tests
(tests/test_gpu_icc.py, test_icc8.py, test_icc16.py) prove the arithmetic; flat launches, the grid cap, offsets beyond 2^31 bytes and
every span index in between exist only at this size -- and these are the frames the profile rows of DESIGN.md section 6.4 time.

  32-bit documents (T2): max |dcode| <= 1, exact >= 99.5 % per plane (harness.T2_MIN_EXACT_ICC, the bar of the small-frame tests);
  8-bit documents (matrix/TRC through lcms2's matrix-shaper, LUT-based through its 33^3 table): every byte of every plane equal.
(16-bit documents: tests/test_icc16.py::test_gpu_full_frame_photograph_bit_exact.)"""
import concurrent.futures
import ctypes
import os
import time

import numpy as np
import pytest

import harness
from test_gpu_fullsize import _device_frame, _oracle_frame

pkg = harness.pkg
pytestmark = pytest.mark.gpu

ICC_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle_icc.so")
W = H = 8192


@pytest.fixture(scope="module")
def lcms():
    if not os.path.exists(ICC_LIB):
        pytest.skip("oracle/liboracle_icc.so not built (lcms2 absent)")
    L = ctypes.CDLL(ICC_LIB)
    L.oracle_icc_make_profile.restype = ctypes.c_int32
    L.oracle_icc_make_profile.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_void_p, ctypes.c_uint32]
    L.oracle_icc_make_a2b_profile.restype = ctypes.c_int32
    L.oracle_icc_make_a2b_profile.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32]
    rows = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    for fn in (L.oracle_icc_convert_rows_to_rec2020, L.oracle_icc_convert_rows_to_srgb_float, L.oracle_icc_convert_rows_to_srgb8):
        fn.restype = ctypes.c_int32
        fn.argtypes = rows
    L.oracle_icc_transform8_open.restype = ctypes.c_void_p
    L.oracle_icc_transform8_open.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    L.oracle_icc_transform16_close.argtypes = [ctypes.c_void_p]
    return L


def _profile(L, kind, trc, g):
    buf = ctypes.create_string_buffer(1 << 16)
    n = L.oracle_icc_make_profile(kind, trc, g, buf, len(buf))
    assert n > 0
    return buf.raw[:n]


def _a2b_profile(L, variant):
    buf = ctypes.create_string_buffer(1 << 18)
    n = L.oracle_icc_make_a2b_profile(variant, buf, len(buf))
    assert n > 0
    return buf.raw[:n]


def _convert_rows(fn, icc, rows):
    """lcms2 over the whole frame, in place, like the reference's row loop -- row blocks on a thread each (every call of the oracle's
    entry points opens its own lcms2 context, profile pair and transform: nothing is shared; ctypes drops the GIL)."""
    t0 = time.perf_counter()
    workers = min(64, os.cpu_count() or 8)
    step = max(16, -(-rows.shape[0] // (workers * 2)))
    width = rows.shape[1] // 3

    def one(r0):
        part = rows[r0:r0 + step]
        return fn(icc, len(icc), 0, part.ctypes.data, width, part.shape[0], rows.strides[0])
    with concurrent.futures.ThreadPoolExecutor(workers) as ex:
        assert all(rc == 0 for rc in ex.map(one, range(0, rows.shape[0], step)))
    print(f"   lcms2: {width}x{rows.shape[0]} in {time.perf_counter() - t0:.2f} s on {workers} threads")


def _gpu_frame(gpu, torch, dev, d, frame, xf):
    bufs, ptrs, strides = {}, [None] * 4, [0] * 4
    ssz = 2 if d.bit_depth > 8 else 1
    for pl, (w, xs, ys) in harness.write_planes(d).items():
        bufs[pl] = torch.zeros(((d.height + ys) >> ys, w * ssz), dtype=torch.uint8, device=dev)
        ptrs[pl], strides[pl] = bufs[pl].data_ptr(), bufs[pl].stride(0)
    gpu.write_rows(d, 0, d.height, frame.data_ptr(), frame.stride(0) * frame.element_size(), ptrs, strides, mem=pkg.MEM_DEVICE,
                   stream=torch.cuda.current_stream(dev).cuda_stream, icc=xf)
    torch.cuda.synchronize(dev)
    return bufs


def _compare(torch, dev, name, d, want, got, exact_bar):
    ssz = 2 if d.bit_depth > 8 else 1
    for pl, (w, xs, ys) in harness.write_planes(d).items():
        h = (d.height + ys) >> ys
        wt = torch.from_numpy(want[pl][:h, :w]).to(dev)
        wt = wt.view(torch.int16) if ssz == 2 else wt
        gt = (got[pl].view(torch.int16) if ssz == 2 else got[pl])[:h, :w]
        if exact_bar is None:
            assert torch.equal(gt, wt), (name, pl)
            continue
        diff = (gt.to(torch.int32) - wt.to(torch.int32)).abs()
        exact = 1.0 - int(torch.count_nonzero(diff)) / diff.numel()
        print(f"{name} plane {pl}: {h}x{w} samples, exact {exact:.6f}, max |dcode| {int(diff.max())}")
        assert int(diff.max()) <= 1, (name, pl)
        assert exact >= exact_bar, (name, pl, exact)


HDR = dict(width=W, height=H, depth=32, planes=3, transfer=pkg.TRANSFER_PQ, peak_nits=80, alpha_state=pkg.ALPHA_NONE,
           matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
F32 = {
    # (profile kind, trc, parameter), sampled?, WriteDesc fields, the kernel the frame must land on
    "linear-P3-doc-default-HDR-save-12bit-422-nearest": ((1, 0, 1.0), False, dict(HDR, bit_depth=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                                                                  chroma_downsampling=pkg.DOWNSAMPLE_NEAREST), ("write_rgb32_ycbcr_sub_hot", "icc=1")),
    "linear-P3-doc-reference-handoff-12bit": ((1, 0, 1.0), False, dict(HDR, bit_depth=12, output=pkg.OUT_REFERENCE), ("write_rgb32_icc1_ycbcr444_hot", "icc=1")),
    "gamma2.2-AdobeRGB-doc-10bit-444": ((3, 0, 2.19921875), False, dict(HDR, bit_depth=10, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444), ("icc=2",)),
    "sampled-curve-P3-doc-10bit-444": ((1, 3, 1024), True, dict(HDR, bit_depth=10, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444), ("icc=6",)),
}


@pytest.mark.parametrize("name", list(F32))
def test_fullsize_32bit_document_behind_a_profile(gpu, lcms, name):
    import torch
    dev = f"cuda:{gpu.device}"
    (kind, trc, g), sampled, kw, kernel = F32[name]
    icc = _profile(lcms, kind, trc, g)
    d = pkg.WriteDesc(**kw)
    frame = _device_frame(torch, dev, d)                  # SURVEY 8d distribution: 10 % highlights up to 12.5, 0.1 % small negatives
    if not (trc == 0 and g == 1.0) and not sampled:
        frame = frame.abs()                               # parametric non-linear curves: stay where every lcms2 build agrees (test_gpu_icc.py)
    xf = gpu.icc_prepare_sampled(icc, pkg.ICC_TARGET_REC2020_LINEAR) if sampled else gpu.icc_prepare(icc)
    got = _gpu_frame(gpu, torch, dev, d, frame, xf)
    label = gpu.last_kernel()
    assert all(k in label for k in kernel), label
    host = frame.cpu().numpy()
    _convert_rows(lcms.oracle_icc_convert_rows_to_rec2020, icc, host)
    want = _oracle_frame(d, host)
    _compare(torch, dev, name, d, want, got, harness.T2_MIN_EXACT_ICC)


def _photograph_like_u8(seed=78):
    """Large-scale gradients + a few codes of noise (the content of tools/bench_configs.py's photograph rows), 8 bit."""
    rng = np.random.default_rng(seed)
    y = np.linspace(0, 1, 1024, dtype=np.float32).reshape(-1, 1)
    x = np.linspace(0, 1, W, dtype=np.float32).reshape(-1, 1)
    ph = np.array([0.0, 2.1, 4.2], dtype=np.float32).reshape(1, 3)
    a = (6.0 * x + ph).reshape(1, -1)
    img = np.sin(a + 3.0 * y) * np.cos(2.0 * y - np.repeat(x, 3, axis=1).reshape(1, -1))
    base = ((0.5 + 0.45 * img) * 236.0 + 8.0).astype(np.int16)
    frame = np.tile(base, (H // 1024, 1))
    frame += rng.integers(-6, 7, size=frame.shape, dtype=np.int8)
    return np.clip(frame, 0, 255).astype(np.uint8)


U8 = {
    "AdobeRGB-matrix-TRC-doc-8bit-420": ("shaper", dict(bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT601), "icc=3"),
    "AdobeRGB-matrix-TRC-doc-12bit-422-nearest": ("shaper", dict(bit_depth=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST,
                                                                 matrix_coefficients=pkg.MATRIX_BT601), "icc=3"),
    "LUT-based-A2B-doc-8bit-420": ("a2b", dict(bit_depth=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT601), "icc=7"),
}


def _clut8_from_lcms(gpu, L, icc):
    """The table the adapter's bridge would hand over: computed from lcms2's own float transform and proven against its 8-bit one
    (avifgpu_icc_clut8_from_transforms; integration/LcmsTableBridge.cpp does the same with the plug-in's transforms)."""
    h = L.oracle_icc_transform8_open(icc, len(icc), 0)
    assert h
    table = pkg.IccClut16()
    rc = gpu.lib.avifgpu_icc_clut8_from_transforms(ctypes.cast(L.oracle_icc_transform16_run_float, ctypes.c_void_p),
                                                   ctypes.cast(L.oracle_icc_transform8_run, ctypes.c_void_p), h, ctypes.byref(table))
    L.oracle_icc_transform16_close(h)
    assert rc == 0, gpu.lib.avifgpu_last_error()
    return table


@pytest.mark.parametrize("name", list(U8))
def test_fullsize_8bit_document_behind_a_profile(gpu, lcms, name):
    import torch
    dev = f"cuda:{gpu.device}"
    how, kw, kernel = U8[name]
    icc = _profile(lcms, 3, 0, 2.19921875) if how == "shaper" else _a2b_profile(lcms, 1)
    xf = gpu.icc_prepare_shaper8(icc) if how == "shaper" else _clut8_from_lcms(gpu, lcms, icc)
    d = pkg.WriteDesc(width=W, height=H, depth=8, planes=3, alpha_state=pkg.ALPHA_NONE, **kw)
    host = _photograph_like_u8()
    frame = torch.from_numpy(host).to(dev)
    got = _gpu_frame(gpu, torch, dev, d, frame, xf)
    assert kernel in gpu.last_kernel(), gpu.last_kernel()
    _convert_rows(lcms.oracle_icc_convert_rows_to_srgb8, icc, host)
    want = _oracle_frame(d, host)
    _compare(torch, dev, name, d, want, got, None)
