"""Fuzz of the FormatRecord-protocol shim and the host-memory pipeline under it: random documents -- every depth,
gray / RGB with and without alpha, odd and tiny geometries up to a few hundred rows -- saved and opened through
avifgpu_host_create_heif_image / avifgpu_host_read_heif_image with a random maxData (whole image, a few rows, one row) on a random number
of bound contexts (1-4 on the visible device: the N-GPU row split of one image), against the oracle's whole-frame conversion.
tests/test_gpu_host_shim.py fixes the protocol on four documents; this one varies what the tile arithmetic, the staging slots and the
in-place plane gather depend on.  Reference loops reproduced: WriteHeifImage.cpp:169-1139, ReadHeifImage.cpp:83-1178."""
import ctypes
import os

import numpy as np
import pytest

import harness
from fake_host import FakeHost

pkg = harness.pkg
H = pkg.host
pytestmark = pytest.mark.gpu
FUZZ_N = int(os.environ.get("AVIFGPU_FUZZ_SHIM_N", "96"))
_bound = {"n": 1}


def _bind(n):
    """n contexts on the visible device(s); re-binding only when the count changes (cases are grouped by it: i % 4)."""
    import torch
    if _bound["n"] != n:
        ndev = torch.cuda.device_count()
        pkg.AvifGpu(devices=[k % ndev for k in range(n)])
        _bound["n"] = n


@pytest.fixture(scope="module", autouse=True)
def _one_context_afterwards():
    yield
    if _bound["n"] != 1:
        pkg.AvifGpu(0)
        _bound["n"] = 1


def _max_data(rng, row_bytes, height):
    return int(rng.choice([0, row_bytes, 2 * row_bytes, 5 * row_bytes + 3, 17 * row_bytes, max(1, height // 3) * row_bytes, 1 << 30, 1]))


def _write_case(i):
    rng = np.random.default_rng(31337 + i)
    depth = int(rng.choice([8, 16, 32]))
    planes = int(rng.choice([1, 2, 3, 3, 4, 4]))
    alpha = pkg.ALPHA_NONE if planes in (1, 3) else int(rng.choice([pkg.ALPHA_STRAIGHT, pkg.ALPHA_PREMULTIPLIED]))
    bits = int(rng.choice({8: [8, 10, 12], 16: [8, 10, 12], 32: [10, 12]}[depth]))
    w = int(rng.choice([1, 2, 3, 7, 8, 64, 0, 0, 0]))
    w = w or int(rng.integers(9, 700))
    h = int(rng.choice([1, 2, 3, 0, 0, 0]))
    h = h or int(rng.integers(4, 260))
    output = pkg.OUT_REFERENCE if planes < 3 or rng.random() < 0.4 else pkg.OUT_YCBCR
    chroma = int(rng.choice([pkg.CHROMA_444, pkg.CHROMA_422, pkg.CHROMA_420]))
    matrix = int(rng.choice([pkg.MATRIX_BT601, pkg.MATRIX_BT709, pkg.MATRIX_BT2020_NCL]))
    kw = dict(width=w, height=h, depth=depth, planes=planes, bit_depth=bits, alpha_state=alpha, output=output, chroma=chroma,
              matrix_coefficients=matrix, color_primaries=pkg.PRIMARIES_BT709, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST)
    if depth == 32:
        kw.update(transfer=pkg.TRANSFER_PQ, peak_nits=int(rng.choice([80, 1000])))
        if alpha == pkg.ALPHA_PREMULTIPLIED:
            kw["alpha_state"] = pkg.ALPHA_STRAIGHT                   # premultiply is disabled for HDR saves (Write.cpp:251-257)
    return kw, _max_data(rng, w * planes * depth // 8, h)


@pytest.mark.parametrize("i", range(FUZZ_N))
def test_shim_save_fuzz(gpu, i):
    kw, max_data = _write_case(i)
    _bind(1 + i % 4)
    d = pkg.WriteDesc(**kw)
    src = harness.make_write_source(d, seed=i)
    host = FakeHost(d.width, d.height, d.depth, d.planes, max_data=max_data, image=src)
    opts = H.SaveUIOptions(imageBitDepth=d.bit_depth, hdrTransferFunction=d.transfer, pq=H.PQOptions(d.peak_nits), chromaSubsampling=d.chroma, lossless=0)
    img = H.Image()
    code = gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host.fr), d.alpha_state, ctypes.byref(opts), d.output, d.matrix_coefficients,
                                                  pkg.PRIMARIES_BT709, ctypes.byref(img))
    assert code == 0, (kw, max_data, gpu.lib.avifgpu_last_error())
    try:
        want = harness.oracle_write(d, src)
        ssz = 2 if d.bit_depth > 8 else 1
        for pl, (w, xs, ys) in harness.write_planes(d).items():
            h = (d.height + ys) >> ys
            raw = (ctypes.c_uint8 * (img.stride[pl] * h)).from_address(img.plane[pl])
            got = np.frombuffer(raw, dtype=np.uint8).reshape(h, img.stride[pl])[:, :w * ssz]
            got = got.view(np.uint16) if ssz == 2 else got
            if d.depth == 32:
                assert np.abs(got.astype(np.int32) - want[pl].astype(np.int32)).max() <= 1, (kw, max_data, pl)
            else:
                assert np.array_equal(got, want[pl]), (kw, max_data, pl, 1 + i % 4)
        # the protocol: ascending full-width rectangles that tile [0, H), one abortProc poll per tile
        assert host.rects[0][0] == 0 and host.rects[-1][2] == d.height
        assert all(a[2] == b[0] for a, b in zip(host.rects[:-1], host.rects[1:])) and all(r[1] == 0 and r[3] == d.width for r in host.rects)
        assert host.polls == len(host.rects)
    finally:
        gpu.lib.avifgpu_image_free(ctypes.byref(img))


def _read_case(i):
    rng = np.random.default_rng(73313 + i)
    bits, depth = [(8, 8), (8, 8), (10, 16), (12, 16), (10, 32), (12, 32)][int(rng.integers(0, 6))]
    cs = int(rng.choice([pkg.COLORSPACE_YCBCR, pkg.COLORSPACE_YCBCR, pkg.COLORSPACE_YCBCR, pkg.COLORSPACE_RGB, pkg.COLORSPACE_MONOCHROME]))
    chroma = {pkg.COLORSPACE_YCBCR: int(rng.choice([pkg.CHROMA_444, pkg.CHROMA_422, pkg.CHROMA_420])),
              pkg.COLORSPACE_RGB: pkg.CHROMA_444, pkg.COLORSPACE_MONOCHROME: pkg.CHROMA_MONOCHROME}[cs]
    w = int(rng.choice([1, 2, 3, 7, 8, 64, 0, 0, 0])) or int(rng.integers(9, 700))
    h = int(rng.choice([1, 2, 3, 0, 0, 0])) or int(rng.integers(4, 260))
    kw = dict(width=w, height=h, colorspace=cs, chroma=chroma, bit_depth=bits, depth=depth,
              alpha_state=int(rng.choice([pkg.ALPHA_NONE, pkg.ALPHA_STRAIGHT, pkg.ALPHA_PREMULTIPLIED])),
              matrix_coefficients=pkg.MATRIX_RGB_GBR if cs == pkg.COLORSPACE_RGB else int(rng.choice([pkg.MATRIX_BT601, pkg.MATRIX_BT709, pkg.MATRIX_BT2020_NCL])),
              color_primaries=pkg.PRIMARIES_BT709, full_range_flag=int(rng.random() < 0.7) if cs != pkg.COLORSPACE_RGB else 1)
    if depth == 32:
        kw.update(transfer_characteristics=pkg.TC_PQ if cs == pkg.COLORSPACE_MONOCHROME else int(rng.choice([pkg.TC_PQ, pkg.TC_HLG, pkg.TC_SMPTE428])),
                  pq_peak_nits=int(rng.choice([80, 1000])), hlg_apply_ootf=int(rng.random() < 0.5), hlg_display_gamma=float(rng.choice([1.0, 1.2, 1.4])),
                  hlg_peak_nits=int(rng.choice([600, 1000])))
    nch = (1 if cs == pkg.COLORSPACE_MONOCHROME else 3) + (0 if kw["alpha_state"] == pkg.ALPHA_NONE else 1)
    return kw, _max_data(rng, w * nch * depth // 8, h)


@pytest.mark.parametrize("i", range(FUZZ_N))
def test_shim_open_fuzz(gpu, i):
    kw, max_data = _read_case(i)
    _bind(1 + i % 4)
    d = pkg.ReadDesc(**kw)
    planes = harness.make_read_source(d, seed=i)
    want = harness.oracle_read(d, planes)
    nch = harness.read_channels(d)
    host = FakeHost(d.width, d.height, d.depth, nch, max_data=max_data)
    img = H.Image(width=d.width, height=d.height, colorspace=d.colorspace, chroma=d.chroma, bit_depth=d.bit_depth)
    for pl, a in planes.items():
        img.plane[pl], img.stride[pl] = a.ctypes.data, a.strides[0]
    nclx = H.Nclx(d.color_primaries, d.transfer_characteristics, d.matrix_coefficients, d.full_range_flag)
    load = H.LoadUIOptions(hlg=H.HLGOptions(d.hlg_apply_ootf, d.hlg_display_gamma, d.hlg_peak_nits), pq=H.PQOptions(d.pq_peak_nits))
    code = gpu.lib.avifgpu_host_read_heif_image(ctypes.byref(img), d.alpha_state, ctypes.byref(nclx), ctypes.byref(load), ctypes.byref(host.fr))
    assert code == 0, (kw, max_data, gpu.lib.avifgpu_last_error())
    if d.depth == 32:
        g, w_ = host.image.astype(np.float64), want.astype(np.float64)
        assert np.all(np.abs(g - w_) <= 1e-4 * np.abs(w_) + 1e-9), (kw, max_data)
    else:
        assert np.array_equal(host.image, want), (kw, max_data, 1 + i % 4)
    assert host.rects[0][0] == 0 and host.rects[-1][2] == d.height
    assert all(a[2] == b[0] for a, b in zip(host.rects[:-1], host.rects[1:]))
