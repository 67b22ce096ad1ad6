"""READ direction at FULL size on one MI355X: what test_gpu_fullsize.py does for the write direction.
 (a) oracle-checked stripes of rows -- top, an interior odd-ish offset, and the bottom edge -- of the frame converted in ONE launch
     (the stripes are run through the CPU oracle as tiles of the full image, so the chroma rows they need are addressed exactly
     like the kernel addresses them);
 (b) row-tile invariance: the frame converted as 8 even-row tiles (the 8-GPU sharding) is byte-identical to the one-launch frame.
Bars: integer hosts bit-exact; f32 hosts the T2 read bar of tests/test_gpu_read.py (|gpu - oracle| <= 1e-4 |oracle| + 1e-9)."""
import ctypes

import numpy as np
import pytest

import harness

pkg = harness.pkg
pytestmark = pytest.mark.gpu

HDR = dict(colorspace=0, depth=32, matrix_coefficients=9, color_primaries=9, transfer_characteristics=16, pq_peak_nits=80)
CONFIGS = {
    "R-8192-10bit-444-pq-f32": dict(width=8192, height=8192, chroma=3, bit_depth=10, alpha_state=0, **HDR),
    "R-8192-10bit-420-pq-f32": dict(width=8192, height=8192, chroma=1, bit_depth=10, alpha_state=0, **HDR),
    "R-8192-12bit-422-pq-f32 (what the plug-in's default HDR save decodes to)": dict(width=8192, height=8192, chroma=2, bit_depth=12, alpha_state=0, **HDR),
    "R-16384-12bit-444-alpha-pq-f32 (C5 read back)": dict(width=16384, height=16384, chroma=3, bit_depth=12, alpha_state=1, **HDR),
    "R-8192-8bit-420-709-rgb8": dict(width=8192, height=8192, colorspace=0, chroma=1, bit_depth=8, depth=8, alpha_state=0, matrix_coefficients=1),
    "R-8191x4097-8bit-422-601-rgb8 (odd geometry)": dict(width=8191, height=4097, colorspace=0, chroma=2, bit_depth=8, depth=8, alpha_state=0, matrix_coefficients=6),
    "R-8192-12bit-420-2020-alpha-premult-rgba16": dict(width=8192, height=8192, colorspace=0, chroma=1, bit_depth=12, depth=16, alpha_state=2,
                                                       matrix_coefficients=9, color_primaries=9),
}


def _device_planes(torch, dev, d, seed=4321):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    ssz = 2 if d.bit_depth > 8 else 1
    maxc = (1 << d.bit_depth) - 1
    out = {}
    for pl, (w, xs, ys) in harness.read_planes(d).items():
        h = (d.height + ys) >> ys
        wp = (w * ssz + 15) // 16 * 16 // ssz                # rows padded to 16 bytes, as libheif allocates planes
        t = torch.randint(0, maxc + 1, (h, wp), generator=g, device=dev, dtype=torch.int32)
        out[pl] = t.to(torch.int16 if ssz == 2 else torch.uint8).contiguous()
    return out


def _run(gpu, torch, dev, d, planes, tiles):
    nch = harness.read_channels(d)
    row_bytes = d.width * nch * (d.depth // 8)
    out = torch.zeros((d.height, (row_bytes + 15) // 16 * 16), dtype=torch.uint8, device=dev)
    for r0, n in tiles:
        if n == 0:
            continue
        ptrs, strides = [None] * 4, [0] * 4
        for pl, (w, xs, ys) in harness.read_planes(d).items():
            ptrs[pl] = planes[pl][r0 >> ys].data_ptr()
            strides[pl] = planes[pl].stride(0) * planes[pl].element_size()
        gpu.read_rows(d, r0, n, ptrs, strides, out[r0].data_ptr(), out.stride(0), mem=pkg.MEM_DEVICE,
                      stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    return out, row_bytes


def _oracle_stripe(d, planes, r0, n):
    """The CPU oracle on rows [r0, r0 + n) as a tile of the full image: hand it host copies of exactly the plane rows the tile touches,
    with the pointers of `row r0 of the tile` like avifgpu_read_rows takes them."""
    import oracle_binding
    L = oracle_binding.load()
    host, ptrs, strides = {}, [None] * 4, [0] * 4
    for pl, (w, xs, ys) in harness.read_planes(d).items():
        first = r0 >> ys
        last = min((r0 + n - 1) >> ys, planes[pl].shape[0] - 1)
        host[pl] = np.ascontiguousarray(planes[pl][first:last + 1].cpu().numpy())
        ptrs[pl], strides[pl] = host[pl].ctypes.data, host[pl].strides[0]
    buf, row_bytes = harness._alloc_read_out(d, n)
    code = L.oracle_read_rows(ctypes.byref(d), r0, n, ctypes.byref(pkg.planes4(ptrs)), ctypes.byref(pkg.strides4(strides)),
                              buf.ctypes.data, buf.strides[0])
    assert code == 0
    return harness._view_read(d, buf, n, row_bytes)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_fullsize_read(gpu, name):
    import torch
    dev = f"cuda:{gpu.device}"
    d = pkg.ReadDesc(**CONFIGS[name])
    planes = _device_planes(torch, dev, d)
    whole, row_bytes = _run(gpu, torch, dev, d, planes, [(0, d.height)])
    tiled, _ = _run(gpu, torch, dev, d, planes, pkg.sharding.all_tiles(d.height, 8))
    assert torch.equal(whole, tiled), name
    del tiled
    mid = d.height // 2 + 6                                   # even: tiles start on chroma-row boundaries (the oracle takes any row0 of a tile)
    for r0, n in ((0, 16), (mid, 16), (d.height - 10 - (d.height - 10) % 2, 10 + (d.height - 10) % 2)):
        want = _oracle_stripe(d, planes, r0, n)
        got = whole[r0:r0 + n, :row_bytes].cpu().numpy().copy().view(harness.src_dtype(d.depth)).reshape(n, -1)
        if d.depth == 32:
            w64, g64 = want.astype(np.float64), got.astype(np.float64)
            assert np.all(np.isfinite(g64)), name
            err = np.abs(g64 - w64)
            assert np.all(err <= 1e-4 * np.abs(w64) + 1e-9), (name, r0, float(np.max(err / np.maximum(np.abs(w64), 1e-30))))
        else:
            assert np.array_equal(got, want), (name, r0)
