"""BASELINE.json configurations at FULL size on one MI355X -- the WHOLE frame against the oracle (round 6; rounds 2-5 compared
three stripes of 42 rows):
 (a) every sample of every plane of the frame converted in ONE launch is compared with the CPU oracle run on the same frame on all host
     cores (oracle_write_image_all_cores: oracle_write_rows over 32-row blocks -- byte-identical to the whole-image call,
     tests/test_oracle_properties.py).  Flat launches, the 32 k-block grid cap, > 4 GiB offsets (C5's source is 4.29 GB) and every span
     index in between exist only at this size.  Integer documents: torch.equal on every plane.  Float documents (T2): max |dcode| <= 1,
     exact >= 99.95 % at 10 bit / 99.9 % at 12 bit per plane, and EVERY mismatching sample is shown to lie within 2e-5 relative of a code
     boundary of the float64 function (for Y/Cb/Cr planes: one of the source samples the code depends on does) --
     reference lines reproduced: WriteHeifImage.cpp:1039-1135 (+ libheif's stage B where the output is Y/Cb/Cr);
 (b) row-tile invariance: the frame converted as 8 even-row tiles (the 8-GPU sharding) is byte-identical to the one-launch frame."""
import hashlib

import numpy as np
import pytest

import harness

pkg = harness.pkg
pytestmark = pytest.mark.gpu


def _device_frame(torch, dev, d, seed=1234):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n = d.height * d.width * d.planes
    if d.depth == 8:
        t = torch.randint(0, 256, (n,), generator=g, device=dev, dtype=torch.uint8)
    elif d.depth == 16:
        t = torch.randint(0, 32769, (n,), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    else:
        t = torch.rand(n, generator=g, device=dev, dtype=torch.float32)
        m = torch.rand(n, generator=g, device=dev, dtype=torch.float32)
        t = torch.where(m < 0.10, 1.0 + 11.5 * t, t)
        t = torch.where(m > 0.999, -0.01 * t, t)
        if d.planes == 4:
            t.view(-1, 4)[:, 3].clamp_(0.0, 1.0)
    return t.view(d.height, d.width * d.planes)


def _run(gpu, torch, dev, d, frame, tiles):
    bufs = {}
    ssz = 2 if d.bit_depth > 8 else 1
    for pl, (w, xs, ys) in harness.write_planes(d).items():
        bufs[pl] = torch.zeros(((d.height + ys) >> ys, w * ssz), dtype=torch.uint8, device=dev)
    esz = frame.element_size()
    for r0, n in tiles:
        if n == 0:
            continue
        ptrs, strides = [None] * 4, [0] * 4
        for pl, (w, xs, ys) in harness.write_planes(d).items():
            ptrs[pl] = bufs[pl][r0 >> ys].data_ptr()
            strides[pl] = bufs[pl].stride(0)
        gpu.write_rows(d, r0, n, frame[r0].data_ptr(), frame.stride(0) * esz, ptrs, strides, mem=pkg.MEM_DEVICE,
                       stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    return bufs


def _digest(torch, bufs):
    h = hashlib.sha256()
    for pl in sorted(bufs):
        # checksum of checksums: 64-bit wraparound sums per plane, computed on the device
        t = bufs[pl].view(torch.int16).to(torch.int64) if bufs[pl].shape[1] % 2 == 0 else bufs[pl].to(torch.int64)
        w = torch.arange(1, t.shape[1] + 1, device=t.device, dtype=torch.int64)
        h.update(str(int((t * w).sum().item())).encode())
        h.update(str(int(t.sum().item())).encode())
    return h.hexdigest()


CONFIGS = {
    "C2-4096-rgb8-420-709": dict(width=4096, height=4096, depth=8, planes=3, bit_depth=8, alpha_state=0,
                                 output=1, chroma=1, matrix_coefficients=1),
    "C3-8192-rgb16-12bit-444-2020": dict(width=8192, height=8192, depth=16, planes=3, bit_depth=12, alpha_state=0,
                                         output=1, chroma=3, matrix_coefficients=9, color_primaries=9),
    "C4-8192-f32-pq-10bit-444": dict(width=8192, height=8192, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80,
                                     alpha_state=0, output=1, chroma=3, matrix_coefficients=9, color_primaries=9),
    "C4-8192-f32-pq-10bit-420": dict(width=8192, height=8192, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80,
                                     alpha_state=0, output=1, chroma=1, matrix_coefficients=9, color_primaries=9),
    "C4-8192-f32-pq-10bit-reference-handoff (what integration/ ships)": dict(width=8192, height=8192, depth=32, planes=3, bit_depth=10,
                                                                             transfer=0, peak_nits=80, alpha_state=0, output=0),
    "D12-8192-f32-pq-12bit-422-nearest (the plug-in's default HDR save)": dict(width=8192, height=8192, depth=32, planes=3, bit_depth=12,
                                                                               transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=2,
                                                                               chroma_downsampling=1, matrix_coefficients=9, color_primaries=9),
    "D8-8192-rgb8-8bit-422-nearest-601 (the plug-in's default SDR save)": dict(width=8192, height=8192, depth=8, planes=3, bit_depth=8,
                                                                               alpha_state=0, output=1, chroma=2, chroma_downsampling=1,
                                                                               matrix_coefficients=6),
    "C5-16384-f32a-pq-12bit-444": dict(width=16384, height=16384, depth=32, planes=4, bit_depth=12, transfer=0,
                                       peak_nits=80, alpha_state=1, output=1, chroma=3, matrix_coefficients=9,
                                       color_primaries=9),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_fullsize(gpu, name):
    import torch
    dev = f"cuda:{gpu.device}"
    d = pkg.WriteDesc(**CONFIGS[name])
    frame = _device_frame(torch, dev, d)
    whole = _run(gpu, torch, dev, d, frame, [(0, d.height)])
    tiled = _run(gpu, torch, dev, d, frame, pkg.sharding.all_tiles(d.height, 8))
    assert _digest(torch, whole) == _digest(torch, tiled), name
    for pl in whole:
        assert torch.equal(whole[pl], tiled[pl]), (name, pl)
    del tiled
    _check_whole_frame(torch, dev, name, d, frame, whole)


M1, M2 = np.float32(2610.0) / np.float32(16384.0), np.float32(2523.0) / np.float32(4096.0) * np.float32(128.0)
C1 = np.float32(3424.0) / np.float32(4096.0)
C2, C3 = np.float32(2413.0) / np.float32(4096.0) * np.float32(32.0), np.float32(2392.0) / np.float32(4096.0) * np.float32(32.0)


def _pq64(x, peak):
    """LinearToPQ (ColorTransfer.cpp:69-92) with the reference's float constants, evaluated in float64 (as tests/test_gpu_t2_truth.py)."""
    x = x.astype(np.float64)
    X = np.power(np.maximum(x, 0.0) * float(np.float32(peak) / np.float32(10000.0)), float(M1))
    return np.where(x < 0, 0.0, np.power((float(C1) + float(C2) * X) / (1.0 + float(C3) * X), float(M2)))


def _oracle_frame(d, host):
    """The whole frame through the oracle on every host core; returns {plane: numpy array with the harness' padded stride}."""
    import ctypes
    import time
    import oracle_binding
    L = oracle_binding.load()
    bufs = harness._alloc_write_out(d, d.height)
    ptrs = pkg.planes4([bufs[i].ctypes.data if i in bufs else None for i in range(4)])
    strides = pkg.strides4([bufs[i].strides[0] if i in bufs else 0 for i in range(4)])
    n = ctypes.c_int32(0)
    t0 = time.perf_counter()
    rc = L.oracle_write_image_all_cores(ctypes.byref(d), host.ctypes.data, host.strides[0], ctypes.byref(ptrs), ctypes.byref(strides), ctypes.byref(n))
    assert rc == 0
    print(f"   oracle: {d.width}x{d.height} on {n.value} threads in {time.perf_counter() - t0:.2f} s")
    return bufs


def _check_whole_frame(torch, dev, name, d, frame, whole):
    host = frame.cpu().numpy()
    if d.depth == 16:
        host = host.view(np.uint16)
    want = _oracle_frame(d, host)
    float_tier = d.depth == 32
    ssz = 2 if d.bit_depth > 8 else 1
    maxv = (1 << d.bit_depth) - 1
    color = 3 if d.planes >= 3 else 1
    for pl, (w, xs, ys) in harness.write_planes(d).items():
        h = (d.height + ys) >> ys
        wt = torch.from_numpy(want[pl][:h, :w]).to(dev)
        wt = wt.view(torch.int16) if ssz == 2 else wt
        gt = whole[pl].view(torch.int16) if ssz == 2 else whole[pl]
        gt = gt[:h, :w]
        if not float_tier:
            assert torch.equal(gt, wt), (name, pl)
            continue
        diff = (gt.to(torch.int32) - wt.to(torch.int32)).abs()
        bad = torch.nonzero(diff)                     # (k, 2): row, sample
        exact = 1.0 - bad.shape[0] / diff.numel()
        print(f"{name} plane {pl}: {h}x{w} samples, exact {exact:.6f}, {bad.shape[0]} mismatches, max |dcode| {int(diff.max())}")
        assert int(diff.max()) <= 1, (name, pl)
        assert exact >= (0.9995 if d.bit_depth == 10 else 0.999), (name, pl, exact)
        if bad.shape[0] == 0:
            continue
        hi_code = torch.maximum(gt[bad[:, 0], bad[:, 1]], wt[bad[:, 0], bad[:, 1]]).cpu().numpy().astype(np.int64)   # the boundary between the two
        bad = bad.cpu().numpy()
        rows, cols = bad[:, 0], bad[:, 1]
        if d.transfer != pkg.TRANSFER_PQ:
            continue
        if d.output == pkg.OUT_REFERENCE and d.planes >= 3:
            ch = cols % d.planes
            assert np.all(ch < color), (name, "alpha must be exact")          # alpha: clamp * max truncated, no curve
            v = host[rows, cols]
            t = np.clip(_pq64(v, d.peak_nits) * maxv, 0, maxv)
            dist = np.abs(t - hi_code)
            assert np.all(dist <= 2e-5 * t + 1e-3), (name, pl, float(dist.max()))
            continue
        assert pl != 3, (name, "alpha plane must be exact")
        # Y / Cb / Cr (or gray Y): the code is a function of the pixel's curve codes (of the block's pixels for sub-sampled chroma):
        # at least one of those source samples sits on a code boundary of the exact function
        worst = np.full(rows.shape, np.inf)
        fx = (1 << xs) if d.chroma_downsampling == pkg.DOWNSAMPLE_AVERAGE else 1
        fy = (1 << ys) if d.chroma_downsampling == pkg.DOWNSAMPLE_AVERAGE else 1
        for dy in range(fy):
            for dx in range(fx):
                sy = np.minimum((rows << ys) + dy, d.height - 1)
                sx = np.minimum((cols << xs) + dx, d.width - 1)
                for c in range(color):
                    v = host[sy, sx * d.planes + c]
                    t = np.clip(_pq64(v, d.peak_nits) * maxv, 0, maxv)
                    dist = np.abs(t - np.rint(t))
                    rel = dist / np.maximum(2e-5 * t + 1e-3, 1e-30)
                    worst = np.minimum(worst, rel)
        assert np.all(worst <= 1.0), (name, pl, float(worst.max()))
