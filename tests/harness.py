"""Shared test plumbing: synthetic inputs (SURVEY.md section 8d distributions, seed 1234), plane geometry,
and thin runners that push the SAME descriptor + buffers through the CPU oracle and through the C-ABI.

The oracle is only ever the checker here.
"""
from __future__ import annotations

import ctypes
import hashlib

import numpy as np

import __graft_entry__ as entry
import oracle_binding

pkg = entry.load_package()

SEED = 1234


def align(v, a):
    return (v + a - 1) // a * a


# ------------------------------------------------------------------------------------------------
# geometry (independent restatement; tests cross-check it against avifgpu_write_plane_geometry)
# ------------------------------------------------------------------------------------------------
def chroma_shift(chroma):
    return {pkg.CHROMA_444: (0, 0), pkg.CHROMA_422: (1, 0), pkg.CHROMA_420: (1, 1)}.get(chroma, (0, 0))


def src_dtype(depth):
    return {8: np.uint8, 16: np.uint16, 32: np.float32}[depth]


def write_planes(desc):
    """{plane_index: (samples_per_row, xshift, yshift)} for the planes the write path produces."""
    color = desc.planes >= 3
    alpha = desc.planes in (2, 4)
    if desc.output == pkg.OUT_REFERENCE:
        if color:
            return {0: (desc.width * desc.planes, 0, 0)}
        out = {0: (desc.width, 0, 0)}
        if alpha:
            out[3] = (desc.width, 0, 0)
        return out
    xs, ys = chroma_shift(desc.chroma)
    cw = (desc.width + xs) >> xs
    out = {0: (desc.width, 0, 0), 1: (cw, xs, ys), 2: (cw, xs, ys)}
    if alpha:
        out[3] = (desc.width, 0, 0)
    return out


def read_planes(desc):
    alpha = desc.alpha_state != pkg.ALPHA_NONE
    if desc.colorspace == pkg.COLORSPACE_MONOCHROME:
        out = {0: (desc.width, 0, 0)}
    elif desc.colorspace == pkg.COLORSPACE_RGB:
        out = {0: (desc.width, 0, 0), 1: (desc.width, 0, 0), 2: (desc.width, 0, 0)}
    else:
        xs, ys = chroma_shift(desc.chroma)
        cw = (desc.width + xs) >> xs
        out = {0: (desc.width, 0, 0), 1: (cw, xs, ys), 2: (cw, xs, ys)}
    if alpha:
        out[3] = (desc.width, 0, 0)
    return out


def read_channels(desc):
    return (1 if desc.colorspace == pkg.COLORSPACE_MONOCHROME else 3) + (1 if desc.alpha_state != pkg.ALPHA_NONE else 0)


# ------------------------------------------------------------------------------------------------
# synthetic inputs
# ------------------------------------------------------------------------------------------------
def make_write_source(desc, seed=SEED, edge_values=True):
    """(height, width*planes) array in FormatRecord layout (interleaved, alpha last)."""
    rng = np.random.default_rng(seed)
    H, W, P = desc.height, desc.width, desc.planes
    has_alpha = P in (2, 4)
    if desc.depth == 8:
        a = rng.integers(0, 256, size=(H, W, P), dtype=np.uint8)
        if has_alpha:
            al = a[..., -1]
            m = rng.random((H, W))
            al[m < 0.05] = 0
            al[m > 0.95] = 255
    elif desc.depth == 16:
        a = rng.integers(0, 32769, size=(H, W, P), dtype=np.uint16)      # Photoshop 16-bit range [0, 32768]
        if has_alpha:
            al = a[..., -1]
            m = rng.random((H, W))
            al[m < 0.05] = 0
            al[m > 0.95] = 32768
    else:
        a = rng.random((H, W, P), dtype=np.float32)                      # ~90 % in [0,1)
        m = rng.random((H, W, P))
        hi = (1.0 + 11.5 * rng.random((H, W, P))).astype(np.float32)     # ~10 % highlights (1, 12.5]
        a = np.where(m < 0.10, hi, a)
        neg = (-0.01 * rng.random((H, W, P))).astype(np.float32)         # ~0.1 % small negatives
        a = np.where(m > 0.999, neg, a).astype(np.float32)
        if has_alpha:
            al = rng.random((H, W), dtype=np.float32)
            m2 = rng.random((H, W))
            al = np.where(m2 < 0.05, 0.0, al)
            al = np.where(m2 > 0.95, 1.0, al)
            al = np.where((m2 > 0.50) & (m2 < 0.51), 1.5, al)            # out of range: must clamp
            al = np.where((m2 > 0.51) & (m2 < 0.52), -0.5, al)
            a[..., -1] = al.astype(np.float32)
        if edge_values and H * W >= 8:
            flat = a.reshape(-1, P)
            specials = [0.0, 1.0, 0.5, 125.0, 1e-9, 1e-4, 12.5, 0.0125]
            for i, v in enumerate(specials):
                flat[i, : (P - 1 if has_alpha else P)] = v
    return np.ascontiguousarray(a.reshape(H, W * P))


def make_read_source(desc, seed=SEED, stride_pad=0):
    """{plane: (rows, stride_samples) array} of u8/u16 codes; 16-bit containers get a few over-range samples."""
    rng = np.random.default_rng(seed)
    maxc = (1 << desc.bit_depth) - 1
    dt = np.uint16 if desc.bit_depth > 8 else np.uint8
    out = {}
    for pl, (w, xs, ys) in read_planes(desc).items():
        h = (desc.height + ys) >> ys
        stride = align(w, 8) + stride_pad
        arr = np.zeros((h, stride), dtype=dt)
        arr[:, :w] = rng.integers(0, maxc + 1, size=(h, w), dtype=np.int64).astype(dt)
        if pl == 3:
            m = rng.random((h, w))
            arr[:, :w][m < 0.05] = 0
            arr[:, :w][m > 0.95] = maxc
        if desc.bit_depth in (10, 12) and desc.colorspace != pkg.COLORSPACE_RGB:
            m = rng.random((h, w))
            arr[:, :w][m > 0.999] = maxc + 7          # exercises std::min(sample, yuvMaxChannel)
        out[pl] = arr
    return out


# ------------------------------------------------------------------------------------------------
# runners
# ------------------------------------------------------------------------------------------------
def _alloc_write_out(desc, nrows, stride_pad=0):
    ssz = 2 if desc.bit_depth > 8 else 1
    dt = np.uint16 if ssz == 2 else np.uint8
    bufs = {}
    for pl, (w, xs, ys) in write_planes(desc).items():
        h = (nrows + ys) >> ys
        stride = align(w * ssz, 16) // ssz + stride_pad
        bufs[pl] = np.full((max(h, 1), stride), 0xA5A5 if ssz == 2 else 0xA5, dtype=dt)
    return bufs


def _trim(desc, bufs, nrows, planes_fn):
    out = {}
    for pl, (w, xs, ys) in planes_fn(desc).items():
        h = (nrows + ys) >> ys
        out[pl] = bufs[pl][:h, :w].copy()
    return out


def oracle_write(desc, src, row0=0, nrows=None, stride_pad=0, return_raw=False):
    L = oracle_binding.load()
    nrows = desc.height - row0 if nrows is None else nrows
    bufs = _alloc_write_out(desc, nrows, stride_pad)
    tile = src[row0:row0 + nrows]
    ptrs = [bufs[i].ctypes.data if i in bufs else None for i in range(4)]
    strides = [bufs[i].strides[0] if i in bufs else 0 for i in range(4)]
    code = L.oracle_write_rows(ctypes.byref(desc), row0, nrows, tile.ctypes.data if nrows else src.ctypes.data,
                               src.strides[0], ctypes.byref(pkg.planes4(ptrs)), ctypes.byref(pkg.strides4(strides)))
    if code != 0:
        raise pkg.AvifGpuError(code, "oracle_write_rows")
    return bufs if return_raw else _trim(desc, bufs, nrows, write_planes)


def gpu_write(gpu, desc, src, row0=0, nrows=None, mem="device", stride_pad=0, return_raw=False):
    nrows = desc.height - row0 if nrows is None else nrows
    bufs = _alloc_write_out(desc, nrows, stride_pad)
    tile = src[row0:row0 + nrows]
    if mem == "host":
        ptrs = [bufs[i].ctypes.data if i in bufs else None for i in range(4)]
        strides = [bufs[i].strides[0] if i in bufs else 0 for i in range(4)]
        gpu.write_rows(desc, row0, nrows, tile.ctypes.data if nrows else src.ctypes.data, src.strides[0], ptrs, strides,
                       mem=pkg.MEM_HOST)
    else:
        import torch
        dev = f"cuda:{gpu.device}"
        d_src = torch.from_numpy(np.ascontiguousarray(tile).view(np.uint8).reshape(-1)).to(dev) if nrows else \
            torch.zeros(16, dtype=torch.uint8, device=dev)
        d_out = {pl: torch.from_numpy(b.view(np.uint8).reshape(-1).copy()).to(dev) for pl, b in bufs.items()}
        ptrs = [d_out[i].data_ptr() if i in d_out else None for i in range(4)]
        strides = [bufs[i].strides[0] if i in bufs else 0 for i in range(4)]
        stream = torch.cuda.current_stream(dev).cuda_stream
        gpu.write_rows(desc, row0, nrows, d_src.data_ptr(), src.strides[0], ptrs, strides, mem=pkg.MEM_DEVICE,
                       stream=stream)
        torch.cuda.synchronize(dev)
        for pl in bufs:
            bufs[pl] = d_out[pl].cpu().numpy().view(bufs[pl].dtype).reshape(bufs[pl].shape)
    return bufs if return_raw else _trim(desc, bufs, nrows, write_planes)


def compare_write(desc, want, got):
    """Max |code difference| and exact-match fraction over all planes."""
    max_abs, total, exact = 0, 0, 0
    for pl in want:
        a = want[pl].astype(np.int64)
        b = got[pl].astype(np.int64)
        assert a.shape == b.shape, (pl, a.shape, b.shape)
        d = np.abs(a - b)
        if d.size:
            max_abs = max(max_abs, int(d.max()))
        total += d.size
        exact += int((d == 0).sum())
    return {"max_abs": max_abs, "exact_frac": exact / max(total, 1), "n": total}


def _alloc_read_out(desc, nrows, pad_bytes=0):
    nch = read_channels(desc)
    row_bytes = desc.width * nch * (desc.depth // 8)
    stride = align(row_bytes, 16) + pad_bytes
    return np.full((max(nrows, 1), stride), 0xA5, dtype=np.uint8), row_bytes


def _view_read(desc, buf, nrows, row_bytes):
    return buf[:nrows, :row_bytes].copy().view(src_dtype(desc.depth)).reshape(nrows, -1)


def _tile_read_ptrs(desc, planes, row0, base_ptr_fn):
    ptrs, strides = [None] * 4, [0] * 4
    for pl, (w, xs, ys) in read_planes(desc).items():
        arr = planes[pl]
        ptrs[pl] = base_ptr_fn(pl) + (row0 >> ys) * arr.strides[0]
        strides[pl] = arr.strides[0]
    return ptrs, strides


def oracle_read(desc, planes, row0=0, nrows=None):
    L = oracle_binding.load()
    nrows = desc.height - row0 if nrows is None else nrows
    buf, row_bytes = _alloc_read_out(desc, nrows)
    ptrs, strides = _tile_read_ptrs(desc, planes, row0, lambda pl: planes[pl].ctypes.data)
    code = L.oracle_read_rows(ctypes.byref(desc), row0, nrows, ctypes.byref(pkg.planes4(ptrs)),
                              ctypes.byref(pkg.strides4(strides)), buf.ctypes.data, buf.strides[0])
    if code != 0:
        raise pkg.AvifGpuError(code, "oracle_read_rows")
    return _view_read(desc, buf, nrows, row_bytes)


def gpu_read(gpu, desc, planes, row0=0, nrows=None, mem="device"):
    nrows = desc.height - row0 if nrows is None else nrows
    buf, row_bytes = _alloc_read_out(desc, nrows)
    if mem == "host":
        ptrs, strides = _tile_read_ptrs(desc, planes, row0, lambda pl: planes[pl].ctypes.data)
        gpu.read_rows(desc, row0, nrows, ptrs, strides, buf.ctypes.data, buf.strides[0], mem=pkg.MEM_HOST)
    else:
        import torch
        dev = f"cuda:{gpu.device}"
        d_pl = {pl: torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(dev) for pl, a in planes.items()}
        d_out = torch.from_numpy(buf.reshape(-1).copy()).to(dev)
        ptrs, strides = _tile_read_ptrs(desc, planes, row0, lambda pl: d_pl[pl].data_ptr())
        stream = torch.cuda.current_stream(dev).cuda_stream
        gpu.read_rows(desc, row0, nrows, ptrs, strides, d_out.data_ptr(), buf.strides[0], mem=pkg.MEM_DEVICE,
                      stream=stream)
        torch.cuda.synchronize(dev)
        buf = d_out.cpu().numpy().reshape(buf.shape)
    return _view_read(desc, buf, nrows, row_bytes)


def digest(arrays):
    h = hashlib.sha256()
    if isinstance(arrays, dict):
        for k in sorted(arrays):
            h.update(np.ascontiguousarray(arrays[k]).tobytes())
    else:
        h.update(np.ascontiguousarray(arrays).tobytes())
    return h.hexdigest()


# ---- float-tier (T2) exact-match bars shared by the GPU tests ----
T2_MIN_EXACT = {10: 0.999, 12: 0.998}          # PQ / HLG / 428 codes against the oracle: what tests/test_gpu_write.py asserts
T2_MIN_EXACT_ICC = 0.995                       # behind a 32-bit ICC stage, against the real lcms2 (measured 99.7-100 %)
T2_SMALL_CASE_MISMATCHES = 4                   # a case of a few thousand samples may hold this many boundary cases whatever the rate


def t2_exact_ok(st, min_exact):
    """The exact-match bar of a float-tier comparison: the rate, or -- on small cases -- a handful of mismatching samples."""
    return st["exact_frac"] >= min_exact or round((1.0 - st["exact_frac"]) * st["n"]) <= T2_SMALL_CASE_MISMATCHES
