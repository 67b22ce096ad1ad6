"""Full-size (BASELINE.json) size-independent properties that involve the READ direction, computed on the device:
 * encode -> decode round trips: the fused write path followed by the read path returns the source within the
   quantisation bound of the intermediate YCbCr codes (C3 integer, C4/C5 through PQ OETF -> EOTF);
 * row-tile invariance of the read path at 8192^2 (8 even-row tiles == one launch, byte for byte);
 * idempotence: converting the same frame twice gives identical bytes (no state leaks between launches)."""
import numpy as np
import pytest

import harness

pkg = harness.pkg
pytestmark = pytest.mark.gpu


def _planes(torch, dev, d):
    ssz = 2 if d.bit_depth > 8 else 1
    return {pl: torch.zeros(((d.height + ys) >> ys, w * ssz), dtype=torch.uint8, device=dev)
            for pl, (w, xs, ys) in harness.write_planes(d).items()}


def _write(gpu, torch, dev, d, frame, bufs, tiles=None):
    esz = frame.element_size()
    for r0, n in (tiles or [(0, d.height)]):
        ptrs, strides = [None] * 4, [0] * 4
        for pl, (w, xs, ys) in harness.write_planes(d).items():
            ptrs[pl], strides[pl] = bufs[pl][r0 >> ys].data_ptr(), bufs[pl].stride(0)
        gpu.write_rows(d, r0, n, frame[r0].data_ptr(), frame.stride(0) * esz, ptrs, strides, mem=pkg.MEM_DEVICE,
                       stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)


def _read(gpu, torch, dev, rd, bufs, out, tiles=None):
    for r0, n in (tiles or [(0, rd.height)]):
        ptrs, strides = [None] * 4, [0] * 4
        for pl, (w, xs, ys) in harness.read_planes(rd).items():
            ptrs[pl], strides[pl] = bufs[pl][r0 >> ys].data_ptr(), bufs[pl].stride(0)
        gpu.read_rows(rd, r0, n, ptrs, strides, out[r0].data_ptr(), out.stride(0) * out.element_size(), mem=pkg.MEM_DEVICE,
                      stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)


def test_c3_roundtrip_rgb16_12bit_444(gpu):
    """C3: 8192^2 RGB16 (0..32768) -> 12-bit BT.2020 4:4:4 -> RGB16.  Error budget on the 32768 scale: 2.45 codes of 4095
    for the YCbCr round trip with libheif's chroma zero point = 19.6, plus half a 12-bit code (4.0)
    for the 32768 -> 4095 rescale of the source: <= 24."""
    import torch
    dev = f"cuda:{gpu.device}"
    W = H = 8192
    d = pkg.WriteDesc(width=W, height=H, depth=16, planes=3, bit_depth=12, alpha_state=0, output=1, chroma=pkg.CHROMA_444,
                      matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
    g = torch.Generator(device=dev); g.manual_seed(3)
    frame = torch.randint(0, 32769, (H, W * 3), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
    bufs = _planes(torch, dev, d)
    _write(gpu, torch, dev, d, frame, bufs)
    again = _planes(torch, dev, d)
    _write(gpu, torch, dev, d, frame, again)
    assert all(torch.equal(bufs[k], again[k]) for k in bufs)                     # idempotent
    rd = pkg.ReadDesc(width=W, height=H, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_444, bit_depth=12, depth=16,
                      alpha_state=0, matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
    out = torch.zeros((H, W * 3), dtype=torch.int16, device=dev)
    _read(gpu, torch, dev, rd, bufs, out)
    tiled = torch.zeros_like(out)
    _read(gpu, torch, dev, rd, bufs, tiled, pkg.sharding.all_tiles(H, 8))
    assert torch.equal(out, tiled)                                               # 8 row tiles == 1 launch
    err = (out.to(torch.int32) & 0xffff) - (frame.to(torch.int32) & 0xffff)
    assert int(err.abs().max()) <= 24, int(err.abs().max())
    assert float(err.float().abs().mean()) < 6.0


@pytest.mark.parametrize("chroma,bits,planes", [(pkg.CHROMA_444, 10, 3), (pkg.CHROMA_444, 12, 4)])
def test_c4_c5_roundtrip_pq(gpu, chroma, bits, planes):
    """C4 / C5-shaped: f32 RGB(A) -> PQ(80 nits) -> 10/12-bit YCbCr(+A) -> PQ EOTF -> f32.  The PQ code step at the top of
    the range (125 x linear at 80 nits) bounds the error: <= 3 codes worth of PQ slope, checked in the PQ domain."""
    import torch
    dev = f"cuda:{gpu.device}"
    W = H = 8192
    alpha = pkg.ALPHA_STRAIGHT if planes == 4 else pkg.ALPHA_NONE
    d = pkg.WriteDesc(width=W, height=H, depth=32, planes=planes, bit_depth=bits, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                      alpha_state=alpha, output=1, chroma=chroma, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                      color_primaries=pkg.PRIMARIES_BT2020)
    g = torch.Generator(device=dev); g.manual_seed(4)
    frame = torch.rand((H, W * planes), generator=g, device=dev, dtype=torch.float32) * 4.0       # up to 4x diffuse white
    bufs = _planes(torch, dev, d)
    _write(gpu, torch, dev, d, frame, bufs, pkg.sharding.all_tiles(H, 8))
    rd = pkg.ReadDesc(width=W, height=H, colorspace=pkg.COLORSPACE_YCBCR, chroma=chroma, bit_depth=bits, depth=32,
                      alpha_state=alpha, matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020,
                      transfer_characteristics=pkg.TC_PQ, pq_peak_nits=80)
    out = torch.zeros((H, W * planes), dtype=torch.float32, device=dev)
    _read(gpu, torch, dev, rd, bufs, out)
    assert bool(torch.isfinite(out).all())
    # compare in the PQ domain (perceptually uniform): |PQ(out) - PQ(src)| <= 3 codes
    def pq(x):
        m1, m2, c1, c2, c3 = 2610 / 16384, 2523 / 4096 * 128, 3424 / 4096, 2413 / 4096 * 32, 2392 / 4096 * 32
        y = (x.clamp(min=0).double() * (80.0 / 10000.0)) ** m1
        return ((c1 + c2 * y) / (1 + c3 * y)) ** m2
    col = slice(None) if planes == 3 else None
    src_c = frame.view(H, W, planes)[..., :3]
    out_c = out.view(H, W, planes)[..., :3]
    maxc = (1 << bits) - 1
    # sample 1/16 of the rows to keep the float64 temporaries small
    e = (pq(out_c[::16]) - pq(src_c[::16])).abs() * maxc
    assert float(e.max()) <= 3.0, float(e.max())
    assert float(e.mean()) < 1.0
    if planes == 4:
        a_src = frame.view(H, W, 4)[..., 3].clamp(0, 1)
        a_out = out.view(H, W, 4)[..., 3]
        assert float((a_out - a_src).abs().max()) <= 1.0 / maxc + 1e-6            # truncating quantiser on alpha
