"""READ direction at FULL size on one MI355X: what test_gpu_fullsize.py does for the write direction.
 (a) the WHOLE frame decoded in ONE launch against the CPU oracle run on the same planes on all host cores (round 6; rounds 3-5 compared
     three stripes of 42 rows): oracle_read_image_all_cores = oracle_read_rows over 32-row blocks, byte-identical to the whole-image call
     (tests/test_oracle_properties.py).  Reference lines the frames reproduce: YuvDecode.cpp:281-696 under ReadHeifImage.cpp:83-400;
 (b) row-tile invariance: the frame converted as 8 even-row tiles (the 8-GPU sharding) is byte-identical to the one-launch frame.
Bars: integer hosts bit-exact on every byte; f32 hosts the T2 read bar of tests/test_gpu_read.py (|gpu - oracle| <= 1e-4 |oracle| + 1e-9)
on every sample."""
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


def _oracle_frame(d, planes, stride):
    """The whole frame through the oracle on all host cores; returns the (height, stride) byte image."""
    import time
    import oracle_binding
    L = oracle_binding.load()
    host, ptrs, strides = {}, [None] * 4, [0] * 4
    for pl in harness.read_planes(d):
        host[pl] = planes[pl].cpu().numpy()
        ptrs[pl], strides[pl] = host[pl].ctypes.data, host[pl].strides[0]
    out = np.empty((d.height, stride), dtype=np.uint8)
    n = ctypes.c_int32(0)
    t0 = time.perf_counter()
    code = L.oracle_read_image_all_cores(ctypes.byref(d), ctypes.byref(pkg.planes4(ptrs)), ctypes.byref(pkg.strides4(strides)),
                                         out.ctypes.data, out.strides[0], ctypes.byref(n))
    assert code == 0
    print(f"   oracle: {d.width}x{d.height} on {n.value} threads in {time.perf_counter() - t0:.2f} s")
    return out


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
    want = torch.from_numpy(_oracle_frame(d, planes, whole.stride(0))).to(dev)
    esz = d.depth // 8
    dt = {8: torch.uint8, 16: torch.int16, 32: torch.float32}[d.depth]
    g = whole.view(dt)[:, :row_bytes // esz]
    w = want.view(dt)[:, :row_bytes // esz]
    if d.depth == 32:
        assert bool(torch.isfinite(g).all()), name
        worst = 0.0
        for r in range(0, d.height, 2048):                  # float64 temporaries of 2048 rows at a time
            g64, w64 = g[r:r + 2048].double(), w[r:r + 2048].double()
            err = (g64 - w64).abs()
            assert bool((err <= 1e-4 * w64.abs() + 1e-9).all()), (name, r)
            worst = max(worst, float((err / w64.abs().clamp_min(1e-6)).max()))
        print(f"{name}: {d.height}x{row_bytes // esz} samples, max relative error against the oracle {worst:.2e}")
    else:
        assert torch.equal(g, w), name
