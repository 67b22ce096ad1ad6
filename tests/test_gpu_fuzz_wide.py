"""Geometry fuzz for the ALIGNED fast paths (streaming kernels, packed 8-bit paths): seeded random widths of a few hundred to a few
thousand pixels, random small heights, every row padded to 16 bytes (what libheif and the library's own staging hand the kernels), a
random even-row tile split.  tests/test_gpu_fuzz.py covers the small / unaligned geometries; this one the ragged last spans, the
one ragged lane of a row, odd heights under 4:2:0 and the tile cuts of the kernels the real documents run on.  Oracle on the same
bytes, same bars as the parity tests.  The size-gated streaming kernels are forced on (tuning-word bit 3)."""
import os

import numpy as np
import pytest

import harness
from test_gpu_u8_fast_path import align

pkg = harness.pkg
pytestmark = pytest.mark.gpu
FUZZ_N = int(os.environ.get("AVIFGPU_FUZZ_N", "160"))      # a soak run: AVIFGPU_FUZZ_N=3000 (3 min on one MI355X)


def gpu_write_padded(gpu, desc, src, row0, nrows):
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
                   stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    raw = {pl: d_out[pl].cpu().numpy().view(bufs[pl].dtype).reshape(bufs[pl].shape) for pl in bufs}
    return harness._trim(desc, raw, nrows, harness.write_planes)


def _case(i):
    rng = np.random.default_rng(4242 + i)
    depth = int(rng.choice([8, 8, 16, 32, 32]))
    planes = int(rng.choice([3, 3, 4, 3, 4, 1, 2]))          # (round 6: gray and gray + alpha documents too -- write_ga_stream / write_int_ref_stream / write_f32_ref_stream)
    chroma = int(rng.choice([pkg.CHROMA_444, pkg.CHROMA_422, pkg.CHROMA_420]))
    choices = {8: [8, 8, 10, 12], 16: [10, 12, 8], 32: [10, 12]}[depth]          # (round 6: 8-bit documents at 12 bit too -- write_rgb8_ycbcr16_hot)
    bits = choices[int(rng.integers(0, len(choices)))]
    w = int(rng.integers(40, 3000))
    w -= w % int(rng.choice([1, 4, 8, 16]))
    w = max(w, 8)
    h = int(rng.integers(1, 10))
    alpha = pkg.ALPHA_NONE if planes in (1, 3) else int(rng.choice([pkg.ALPHA_STRAIGHT, pkg.ALPHA_PREMULTIPLIED]))
    kw = dict(width=w, height=h, depth=depth, planes=planes, bit_depth=bits, alpha_state=alpha, output=pkg.OUT_YCBCR, chroma=chroma,
              matrix_coefficients=int(rng.choice([pkg.MATRIX_BT601, pkg.MATRIX_BT709, pkg.MATRIX_BT2020_NCL])),
              color_primaries=pkg.PRIMARIES_BT709)
    if rng.random() < 0.4:
        kw["chroma_downsampling"] = pkg.DOWNSAMPLE_NEAREST
    if depth == 32:
        kw.update(transfer=int(rng.choice([pkg.TRANSFER_PQ, pkg.TRANSFER_CLIP, pkg.TRANSFER_SMPTE428 if planes == 3 and bits == 12 else pkg.TRANSFER_PQ])),
                  peak_nits=int(rng.choice([80, 1000])))
        if alpha == pkg.ALPHA_PREMULTIPLIED and kw["transfer"] != pkg.TRANSFER_CLIP:
            kw["alpha_state"] = pkg.ALPHA_STRAIGHT               # premultiply is disabled for HDR saves (Write.cpp:251-257)
    cut = 2 * int(rng.integers(0, h // 2 + 1))
    if planes < 3:                                               # gray: Y (+ alpha) planes, the reference's own hand-off (WriteHeifImage.cpp:169-631)
        kw["output"] = pkg.OUT_REFERENCE
        if depth == 32 and kw["transfer"] == pkg.TRANSFER_SMPTE428:
            kw["transfer"] = pkg.TRANSFER_PQ                     # gray documents: PQ or Clip only (:581-582)
    if rng.random() < 0.2:                                       # round 6: the reference's own interleaved hand-off as well (the 8-bit identity case is a copy kernel)
        kw["output"] = pkg.OUT_REFERENCE
    return kw, cut


@pytest.mark.parametrize("i", range(FUZZ_N))
def test_write_fuzz_wide_aligned(gpu, i):
    kw, cut = _case(i)
    d = pkg.WriteDesc(**kw)
    src = harness.make_write_source(d, seed=i)
    float_tier = d.depth == 32
    try:
        gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4 | 8)
        for row0, nrows in ((0, cut), (cut, d.height - cut)):
            if nrows == 0:
                continue
            want = harness.oracle_write(d, src, row0, nrows)
            got = gpu_write_padded(gpu, d, src, row0, nrows)
            st = harness.compare_write(d, want, got)
            print("fuzz-wide kernel:", gpu.last_kernel().split('<')[0] + (' icc' if 'icc=' in gpu.last_kernel() else ''))
            if float_tier:
                assert st["max_abs"] <= 1, (kw, row0, nrows, gpu.last_kernel(), st)
                assert st["exact_frac"] >= 0.98 or st["n"] < 20000, (kw, gpu.last_kernel(), st)
            else:
                assert st["max_abs"] == 0, (kw, row0, nrows, gpu.last_kernel(), st)
    finally:
        gpu.lib.avifgpu_set_hot_variant(1 | 2 | 4)


def _read_case(i):
    rng = np.random.default_rng(9191 + i)
    bits, depth = [(8, 8), (8, 8), (10, 16), (12, 16), (10, 32), (12, 32)][int(rng.integers(0, 6))]
    cs = int(rng.choice([pkg.COLORSPACE_YCBCR, pkg.COLORSPACE_YCBCR, pkg.COLORSPACE_YCBCR, pkg.COLORSPACE_RGB, pkg.COLORSPACE_MONOCHROME]))
    chroma = {pkg.COLORSPACE_YCBCR: int(rng.choice([pkg.CHROMA_444, pkg.CHROMA_422, pkg.CHROMA_420])),
              pkg.COLORSPACE_RGB: pkg.CHROMA_444, pkg.COLORSPACE_MONOCHROME: pkg.CHROMA_MONOCHROME}[cs]
    w = max(8, int(rng.integers(40, 3000)) // int(rng.choice([1, 2, 8, 16])) * int(rng.choice([1, 2, 8, 16])))
    kw = dict(width=w, height=int(rng.integers(1, 10)), colorspace=cs, chroma=chroma, bit_depth=bits, depth=depth,
              alpha_state=int(rng.choice([pkg.ALPHA_NONE, pkg.ALPHA_STRAIGHT, pkg.ALPHA_PREMULTIPLIED])),
              matrix_coefficients=pkg.MATRIX_RGB_GBR if cs == pkg.COLORSPACE_RGB else int(rng.choice([pkg.MATRIX_BT601, pkg.MATRIX_BT709, pkg.MATRIX_BT2020_NCL])),
              color_primaries=pkg.PRIMARIES_BT709, full_range_flag=int(rng.random() < 0.7) if cs != pkg.COLORSPACE_RGB else 1)
    if depth == 32:
        kw.update(transfer_characteristics=pkg.TC_PQ if cs == pkg.COLORSPACE_MONOCHROME else int(rng.choice([pkg.TC_PQ, pkg.TC_HLG, pkg.TC_SMPTE428])),
                  pq_peak_nits=int(rng.choice([80, 1000])))
    return kw


@pytest.mark.parametrize("i", range(FUZZ_N))
def test_read_fuzz_wide_aligned(gpu, i):
    import cases
    from test_gpu_read import _check
    kw = _read_case(i)
    d = pkg.ReadDesc(**kw)
    planes = harness.make_read_source(d, seed=i)
    fixed = {}
    for pl, arr in planes.items():                                 # 16-byte strides, as libheif pads its planes
        out = np.zeros((arr.shape[0], align(arr.shape[1] * arr.itemsize, 16) // arr.itemsize), dtype=arr.dtype)
        out[:, :arr.shape[1]] = arr
        fixed[pl] = out
    cut = 2 * int(np.random.default_rng(i).integers(0, d.height // 2 + 1))
    for row0, nrows in ((0, cut), (cut, d.height - cut)):
        if nrows == 0:
            continue
        want = harness.oracle_read(d, fixed, row0, nrows)
        got = harness.gpu_read(gpu, d, fixed, row0, nrows, mem="device")
        assert "aligned=1" in gpu.last_kernel(), gpu.last_kernel()
        _check(f"read-fuzz-wide-{i}", kw, got, want)
