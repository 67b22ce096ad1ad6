"""Descriptor fuzz -- the ERROR behaviour of the boundary: avifgpu_write_desc / avifgpu_read_desc with every field drawn from a pool of
valid AND invalid values (out-of-range enums, zero / negative sizes, depth / bit-depth / plane / alpha combinations the reference rejects
at WriteHeifImage.cpp:41-85, Write.cpp:303-336, ReadHeifImage.cpp:52-135,418-621), handed to the C-ABI with device buffers that are large
enough for anything the library could accept.  Per descriptor:
  * the call returns (no crash, no hang) with 0 or an OSErr and, on an error, a message in avifgpu_last_error();
  * what the ORACLE rejects the library rejects (it is never more lenient than the restatement of the reference's checks);
  * what both accept comes out equal: bit-exact for integer documents, the T2 bars for 32-bit ones;
  * what only the library rejects must name one of its documented restrictions (LIBRARY_ONLY below).
tests/test_gpu_fuzz*.py draw VALID geometries; this one draws the descriptors themselves."""
import ctypes
import os

import numpy as np
import pytest

import harness

pkg = harness.pkg
pytestmark = pytest.mark.gpu
FUZZ_N = int(os.environ.get("AVIFGPU_FUZZ_DESC_N", "600"))
ROW, ROWS = 2048, 40                       # every buffer: 40 rows of 2048 bytes -- more than any accepted descriptor below can touch

# messages of restrictions that are the library's own (include/avifgpu.h), not the reference's: the oracle has no such check
LIBRARY_ONLY = (b"limited-range output", b"pq_evaluation", b"chroma_zero_point", b"chroma_downsampling", b"nominalPeakBrightness", b"32-bit documents save as 10 or 12 bit",
                b"alpha_state", b"Unsupported color transfer function", b"hlg", b"HLG", b"peak")


def _buffers(torch, dev, n):
    return [torch.zeros((ROWS, ROW), dtype=torch.uint8, device=dev) for _ in range(n)]


WRITE_POOL = dict(width=[0, -1, 1, 2, 3, 7, 8, 17, 31, 40], height=[0, -3, 1, 2, 3, 5, 8, 9], depth=[8, 16, 32, 0, 7, 24, 64], planes=[1, 2, 3, 4, 0, 5, -1],
                  bit_depth=[8, 10, 12, 0, 9, 16], alpha_state=[0, 1, 2, 3, -1], output=[0, 1, 2, -1], chroma=[1, 2, 3, 0, 4, 99], transfer=[0, 1, 2, 3, 4, -1],
                  peak_nits=[80, 1000, 10000, 1, 0, -5, 20000], matrix_coefficients=[1, 5, 6, 9, 0, 2, 14, 255], color_primaries=[1, 9, 12, 2, 0, 255],
                  chroma_downsampling=[0, 1, 2, -1])


def _mutate(rng, kw, pool):
    """A valid descriptor with 0-3 of its fields redrawn from the whole pool (valid and invalid values alike)."""
    for _ in range(int(rng.integers(0, 4))):
        k = list(pool)[int(rng.integers(0, len(pool)))]
        kw[k] = int(pool[k][int(rng.integers(0, len(pool[k])))])
    return kw


def _write_kw(rng):
    pick = lambda pool: int(pool[int(rng.integers(0, len(pool)))])
    depth, planes = pick([8, 16, 32]), pick([1, 2, 3, 3, 4])
    kw = dict(width=pick([1, 2, 3, 7, 8, 17, 31, 40]), height=pick([1, 2, 3, 5, 8, 9]), depth=depth, planes=planes,
              bit_depth=pick({8: [8, 10, 12], 16: [8, 10, 12], 32: [10, 12]}[depth]), alpha_state=pick([1, 2]) if planes in (2, 4) else 0,
              output=pick([0, 1]) if planes >= 3 else 0, chroma=pick([1, 2, 3]), transfer=pick([0, 3]) if depth == 32 else 3,
              peak_nits=pick([80, 1000]), matrix_coefficients=pick([1, 6, 9]), color_primaries=pick([1, 9]), chroma_downsampling=pick([0, 1]))
    return _mutate(rng, kw, WRITE_POOL)


@pytest.mark.parametrize("i", range(FUZZ_N))
def test_write_descriptor_fuzz(gpu, i):
    import torch
    import oracle_binding
    dev = f"cuda:{gpu.device}"
    rng = np.random.default_rng(77000 + i)
    kw = _write_kw(rng)
    d = pkg.WriteDesc(**kw)
    host_src = rng.integers(0, 256, size=(ROWS, ROW), dtype=np.uint8)
    if kw["depth"] == 16:
        host_src.view(np.uint16)[:] = rng.integers(0, 32769, size=(ROWS, ROW // 2), dtype=np.uint16)
    elif kw["depth"] == 32:
        host_src.view(np.float32)[:] = (rng.random((ROWS, ROW // 4), dtype=np.float32) * 1.5 - 0.1)
    nrows = max(kw["height"], 0)
    # the oracle's verdict (CPU, same descriptor, same bytes)
    L = oracle_binding.load()
    o_out = [np.zeros((ROWS, ROW), dtype=np.uint8) for _ in range(4)]
    o_code = L.oracle_write_rows(ctypes.byref(d), 0, nrows, host_src.ctypes.data, ROW,
                                 ctypes.byref(pkg.planes4([a.ctypes.data for a in o_out])), ctypes.byref(pkg.strides4([ROW] * 4)))
    # the library's
    src = torch.from_numpy(host_src).to(dev)
    out = _buffers(torch, dev, 4)
    code = gpu.lib.avifgpu_write_rows(ctypes.byref(d), 0, nrows, src.data_ptr(), ROW, ctypes.byref(pkg.planes4([t.data_ptr() for t in out])),
                                      ctypes.byref(pkg.strides4([ROW] * 4)), pkg.MEM_DEVICE, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    msg = gpu.lib.avifgpu_last_error()
    if code != 0:
        assert msg, (kw, code)
        if o_code == 0 and nrows > 0:
            assert any(t in msg for t in LIBRARY_ONLY), ("the library rejects what the oracle accepts", kw, code, msg)
        return
    assert o_code == 0 or nrows == 0, ("the library accepts what the oracle rejects", kw, o_code)
    if nrows == 0:
        return
    for pl, (w, xs, ys) in harness.write_planes(d).items():
        h = (d.height + ys) >> ys
        nb = w * (2 if d.bit_depth > 8 else 1)
        got, want = out[pl].cpu().numpy()[:h, :nb], o_out[pl][:h, :nb]
        if d.depth != 32:
            assert np.array_equal(got, want), (kw, pl)
        else:
            g, w_ = (got.view(np.uint16), want.view(np.uint16)) if d.bit_depth > 8 else (got, want)
            assert np.abs(g.astype(np.int32) - w_.astype(np.int32)).max() <= 1, (kw, pl)


READ_POOL = dict(width=[0, -1, 1, 2, 3, 7, 8, 17, 31, 40], height=[0, -3, 1, 2, 3, 5, 8, 9], colorspace=[0, 1, 2, 3, -1, 99], chroma=[0, 1, 2, 3, 4, 99],
                 bit_depth=[8, 10, 12, 0, 9, 16], depth=[8, 16, 32, 0, 24], alpha_state=[0, 1, 2, 3, -1], matrix_coefficients=[0, 1, 5, 6, 9, 2, 14, 255],
                 color_primaries=[1, 9, 12, 2, 0, 255], transfer_characteristics=[16, 17, 18, 13, 1, 2, 0, 255], full_range_flag=[0, 1], has_nclx=[0, 1],
                 pq_peak_nits=[80, 1000, 10000, 0, -5])


def _read_kw(rng):
    pick = lambda pool: int(pool[int(rng.integers(0, len(pool)))])
    cs = pick([pkg.COLORSPACE_YCBCR, pkg.COLORSPACE_YCBCR, pkg.COLORSPACE_RGB, pkg.COLORSPACE_MONOCHROME])
    bits, depth = [(8, 8), (10, 16), (12, 16), (10, 32), (12, 32)][int(rng.integers(0, 5))]
    kw = dict(width=pick([1, 2, 3, 7, 8, 17, 31, 40]), height=pick([1, 2, 3, 5, 8, 9]), colorspace=cs,
              chroma={pkg.COLORSPACE_YCBCR: pick([1, 2, 3]), pkg.COLORSPACE_RGB: pkg.CHROMA_444, pkg.COLORSPACE_MONOCHROME: pkg.CHROMA_MONOCHROME}[cs],
              bit_depth=bits, depth=depth, alpha_state=pick([0, 0, 1, 2]),
              matrix_coefficients=pkg.MATRIX_RGB_GBR if cs == pkg.COLORSPACE_RGB else pick([1, 6, 9]), color_primaries=pick([1, 9]),
              transfer_characteristics=pick([16, 17, 18]) if depth == 32 and cs != pkg.COLORSPACE_MONOCHROME else 16, full_range_flag=pick([0, 1, 1]) if cs != pkg.COLORSPACE_RGB else 1,
              has_nclx=1, pq_peak_nits=pick([80, 1000]))
    return _mutate(rng, kw, READ_POOL)


@pytest.mark.parametrize("i", range(FUZZ_N))
def test_read_descriptor_fuzz(gpu, i):
    import torch
    import oracle_binding
    dev = f"cuda:{gpu.device}"
    rng = np.random.default_rng(88000 + i)
    kw = _read_kw(rng)
    d = pkg.ReadDesc(**kw)
    nrows = max(kw["height"], 0)
    host_planes = []
    for _ in range(4):
        a = np.zeros((ROWS, ROW), dtype=np.uint8)
        if kw["bit_depth"] > 8:
            a.view(np.uint16)[:] = rng.integers(0, 1 << min(max(kw["bit_depth"], 1), 12), size=(ROWS, ROW // 2), dtype=np.uint16)
        else:
            a[:] = rng.integers(0, 256, size=(ROWS, ROW), dtype=np.uint8)
        host_planes.append(a)
    L = oracle_binding.load()
    o_out = np.zeros((ROWS, ROW), dtype=np.uint8)
    o_code = L.oracle_read_rows(ctypes.byref(d), 0, nrows, ctypes.byref(pkg.planes4([a.ctypes.data for a in host_planes])),
                                ctypes.byref(pkg.strides4([ROW] * 4)), o_out.ctypes.data, ROW)
    planes = [torch.from_numpy(a).to(dev) for a in host_planes]
    out = torch.zeros((ROWS, ROW), dtype=torch.uint8, device=dev)
    code = gpu.lib.avifgpu_read_rows(ctypes.byref(d), 0, nrows, ctypes.byref(pkg.planes4([t.data_ptr() for t in planes])),
                                     ctypes.byref(pkg.strides4([ROW] * 4)), out.data_ptr(), ROW, pkg.MEM_DEVICE,
                                     torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    msg = gpu.lib.avifgpu_last_error()
    if code != 0:
        assert msg, (kw, code)
        if o_code == 0 and nrows > 0:
            assert any(t in msg for t in LIBRARY_ONLY), ("the library rejects what the oracle accepts", kw, code, msg)
        return
    assert o_code == 0 or nrows == 0, ("the library accepts what the oracle rejects", kw, o_code)
    if nrows == 0:
        return
    nb = d.width * harness.read_channels(d) * (d.depth // 8)
    got, want = out.cpu().numpy()[:nrows, :nb], o_out[:nrows, :nb]
    if d.depth != 32:
        assert np.array_equal(got, want), kw
    else:
        g, w_ = got.view(np.float32).astype(np.float64), want.view(np.float32).astype(np.float64)
        assert np.all(np.abs(g - w_) <= 1e-4 * np.abs(w_) + 1e-9), kw


def test_buffer_arguments_are_checked_before_anything_is_launched(gpu):
    """The pointers and pitches of a call (plain C arguments: the reference has C++ objects there, so these checks are the library's own):
    a null source / destination, a null plane the descriptor needs, a pitch below the row -- formatBadParameters and a message, nothing
    written; planes the descriptor does not produce may be null."""
    import torch
    dev = f"cuda:{gpu.device}"
    stream = torch.cuda.current_stream(dev).cuda_stream
    d = pkg.WriteDesc(width=32, height=8, depth=16, planes=3, bit_depth=10, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                      matrix_coefficients=pkg.MATRIX_BT601)
    src = torch.zeros((8, 32 * 3), dtype=torch.int16, device=dev)
    planes = [torch.full((8, 64), 0x5a, dtype=torch.uint8, device=dev) for _ in range(3)]

    def write(src_ptr, src_pitch, ptrs, pitches):
        code = gpu.lib.avifgpu_write_rows(ctypes.byref(d), 0, 8, src_ptr, src_pitch, ctypes.byref(pkg.planes4(ptrs)), ctypes.byref(pkg.strides4(pitches)),
                                          pkg.MEM_DEVICE, stream)
        torch.cuda.synchronize(dev)
        return code, gpu.lib.avifgpu_last_error()
    good_ptrs, good_pitches = [t.data_ptr() for t in planes] + [None], [64, 64, 64, 0]
    assert write(src.data_ptr(), 32 * 6, good_ptrs, good_pitches)[0] == 0                      # the alpha plane is not produced: null is fine
    for t in planes:
        t.fill_(0x5a)
    bad = [(None, 32 * 6, good_ptrs, good_pitches, b"null buffer"),
           (src.data_ptr(), 32 * 6 - 2, good_ptrs, good_pitches, b"src_row_bytes"),
           (src.data_ptr(), 32 * 6, [good_ptrs[0], None, good_ptrs[2], None], good_pitches, b"destination plane 1 is null"),
           (src.data_ptr(), 32 * 6, good_ptrs, [64, 31, 64, 0], b"dst_stride[1]"),
           (src.data_ptr(), 32 * 6, good_ptrs, [63, 64, 64, 0], b"dst_stride[0]"),
           (src.data_ptr(), 32 * 6, good_ptrs, [64, 64, -64, 0], b"dst_stride[2]")]
    for src_ptr, pitch, ptrs, pitches, text in bad:
        code, msg = write(src_ptr, pitch, ptrs, pitches)
        assert code == pkg.formatBadParameters and text in msg, (text, code, msg)
    assert all(bool((t == 0x5a).all()) for t in planes), "a rejected call must not write"
    code = gpu.lib.avifgpu_write_rows(ctypes.byref(d), 0, 8, src.data_ptr(), 32 * 6, ctypes.byref(pkg.planes4(good_ptrs)), ctypes.byref(pkg.strides4(good_pitches)), 7, stream)
    assert code == pkg.formatBadParameters, "unknown mem_kind"

    r = pkg.ReadDesc(width=32, height=8, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_422, bit_depth=8, depth=8, alpha_state=pkg.ALPHA_NONE,
                     matrix_coefficients=pkg.MATRIX_BT601)
    rp = [torch.zeros((8, 32), dtype=torch.uint8, device=dev) for _ in range(3)]
    out = torch.full((8, 96), 0x5a, dtype=torch.uint8, device=dev)

    def read(ptrs, pitches, dst, dst_pitch):
        code = gpu.lib.avifgpu_read_rows(ctypes.byref(r), 0, 8, ctypes.byref(pkg.planes4(ptrs)), ctypes.byref(pkg.strides4(pitches)), dst, dst_pitch,
                                         pkg.MEM_DEVICE, stream)
        torch.cuda.synchronize(dev)
        return code, gpu.lib.avifgpu_last_error()
    gp, gs = [t.data_ptr() for t in rp] + [None], [32, 32, 32, 0]
    assert read(gp, gs, out.data_ptr(), 96)[0] == 0
    out.fill_(0x5a)
    for ptrs, pitches, dst, dpitch, text in [(gp, gs, None, 96, b"null buffer"), (gp, gs, out.data_ptr(), 95, b"dst_row_bytes too small"),
                                              ([gp[0], gp[1], None, None], gs, out.data_ptr(), 96, b"source plane 2 is null"),
                                              (gp, [31, 32, 32, 0], out.data_ptr(), 96, b"src_stride[0] too small"),
                                              (gp, [32, 15, 32, 0], out.data_ptr(), 96, b"src_stride[1] too small")]:
        code, msg = read(ptrs, pitches, dst, dpitch)
        assert code == pkg.formatBadParameters and text in msg, (text, code, msg)
    assert bool((out == 0x5a).all()), "a rejected call must not write"
