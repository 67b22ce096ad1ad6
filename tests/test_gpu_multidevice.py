"""The in-process multi-GPU row-tile scheduler (avifgpu_init_devices, csrc/pipeline.hip): N bound contexts convert ONE image
from ONE calling thread -- contiguous even-row tiles, one per context, no exchange step (the loops it replaces:
WriteHeifImage.cpp:1017-1029, ReadHeifImage.cpp:141-160, buffer set-up Write.cpp:279-299).  The planes must be byte-identical
for every N.  With fewer GPUs than contexts an ordinal is bound several times (N workers, N stream sets on one device): the
scheduling, slot rotation, gather offsets and error draining are the same code either way."""
import ctypes

import numpy as np
import pytest

import harness
from fake_host import FakeHost

pkg = harness.pkg
H = pkg.host
pytestmark = pytest.mark.gpu


def _bind(n):
    import torch
    ndev = max(torch.cuda.device_count(), 1)
    return pkg.AvifGpu(devices=[i % ndev for i in range(n)])


@pytest.fixture
def rebind():
    yield _bind
    pkg.AvifGpu(0)                      # the rest of the suite runs on one context


WRITE = {
    "c4-444": dict(width=520, height=301, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                   alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                   matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020),
    "c5-420-near": dict(width=512, height=255, depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=1000,
                        alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420,
                        matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020,
                        chroma_downsampling=pkg.DOWNSAMPLE_NEAREST),
    "rgba16-ref": dict(width=333, height=97, depth=16, planes=4, bit_depth=10, alpha_state=pkg.ALPHA_PREMULTIPLIED,
                       output=pkg.OUT_REFERENCE),
    "c2-420": dict(width=1000, height=222, depth=8, planes=3, bit_depth=8, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR,
                   chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT709),
    "gray8a": dict(width=77, height=33, depth=8, planes=2, bit_depth=12, alpha_state=pkg.ALPHA_STRAIGHT,
                   output=pkg.OUT_REFERENCE),
}


@pytest.mark.parametrize("name", list(WRITE))
def test_write_rows_host_n_contexts_equal_one(rebind, monkeypatch, name):
    """avifgpu_write_rows(MEM_HOST): 1, 2, 4, 8 contexts, small sub-tiles so every slot rotates -- same bytes, padding intact."""
    d = pkg.WriteDesc(**WRITE[name])
    src = harness.make_write_source(d, seed=11)
    want = harness.oracle_write(d, src, stride_pad=8, return_raw=True)
    ref = None
    for n, chunk_mb in ((1, 4096), (2, 1), (4, 1), (8, 1), (3, 1)):
        monkeypatch.setenv("AVIFGPU_CHUNK_MB", str(chunk_mb))
        gpu = rebind(n)
        assert gpu.lib.avifgpu_device_count() == n
        got = harness.gpu_write(gpu, d, src, mem="host", stride_pad=8, return_raw=True)
        assert "write" in gpu.last_kernel()
        if ref is None:
            ref = got
            if d.depth != 32 or d.transfer == pkg.TRANSFER_CLIP:
                for pl in want:
                    assert np.array_equal(got[pl], want[pl]), (name, pl)
            else:
                trim = lambda b: harness._trim(d, b, d.height, harness.write_planes)
                st = harness.compare_write(d, trim(want), trim(got))
                assert st["max_abs"] <= 1 and harness.t2_exact_ok(st, harness.T2_MIN_EXACT[d.bit_depth]), st
        for pl in ref:
            assert np.array_equal(got[pl], ref[pl]), (name, n, pl)       # byte-identical for every N, padding included


@pytest.mark.parametrize("depth,lanes,slots", [(0, 2, 4), (1, 1, 4), (1, 2, 2), (1, 4, 3), (2, 2, 2), (4, 3, 8)])
def test_upload_order_settings_give_the_same_bytes(rebind, monkeypatch, depth, lanes, slots):
    """The scheduler's knobs -- uploads of a device unordered / in queue order, 1..4 at a time, lanes x slots below and above the
    4 hardware queues -- change when copies run, never what they carry: many small sub-tiles on 1 and 3 contexts, both
    directions, equal to the defaults' bytes; and a changed AVIFGPU_UPLOAD_DEPTH re-binds like the other knobs."""
    wd = pkg.WriteDesc(**WRITE["c5-420-near"])
    src = harness.make_write_source(wd, seed=21)
    rd = pkg.ReadDesc(width=264, height=151, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_420, bit_depth=10, depth=16,
                      alpha_state=pkg.ALPHA_PREMULTIPLIED, matrix_coefficients=pkg.MATRIX_BT709)
    planes = harness.make_read_source(rd, seed=5, stride_pad=8)
    monkeypatch.setenv("AVIFGPU_CHUNK_MB", "1")
    gpu = rebind(1)
    want_w = harness.gpu_write(gpu, wd, src, mem="host", stride_pad=8, return_raw=True)
    want_r = harness.gpu_read(gpu, rd, planes, mem="host")
    monkeypatch.setenv("AVIFGPU_UPLOAD_DEPTH", str(depth))
    monkeypatch.setenv("AVIFGPU_LANES", str(lanes))
    monkeypatch.setenv("AVIFGPU_SLOTS", str(slots))
    for n in (1, 3):
        gpu = rebind(n)
        for _ in range(2):                                     # twice: the second pass starts from the first one's event ring
            got = harness.gpu_write(gpu, wd, src, mem="host", stride_pad=8, return_raw=True)
            for pl in want_w:
                assert np.array_equal(got[pl], want_w[pl]), (depth, lanes, slots, n, pl)
            assert np.array_equal(harness.gpu_read(gpu, rd, planes, mem="host").view(np.uint8), want_r.view(np.uint8))
    import torch
    ndev = max(torch.cuda.device_count(), 1)
    assert gpu.topology()[0]["workers"] == lanes * len([i for i in range(3) if i % ndev == 0])     # the knob took effect


def test_read_rows_host_n_contexts_equal_one(rebind, monkeypatch):
    for kw in (dict(width=264, height=151, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_420, bit_depth=10, depth=16,
                    alpha_state=pkg.ALPHA_PREMULTIPLIED, matrix_coefficients=pkg.MATRIX_BT709),
               dict(width=200, height=90, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_444, bit_depth=12, depth=32,
                    alpha_state=pkg.ALPHA_NONE, matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020,
                    transfer_characteristics=pkg.TC_PQ, pq_peak_nits=1000),
               dict(width=123, height=45, colorspace=pkg.COLORSPACE_MONOCHROME, chroma=pkg.CHROMA_MONOCHROME, bit_depth=8,
                    depth=8, alpha_state=pkg.ALPHA_STRAIGHT)):
        d = pkg.ReadDesc(**kw)
        planes = harness.make_read_source(d, seed=3, stride_pad=8)
        ref = None
        for n, chunk_mb in ((1, 4096), (2, 1), (4, 1), (8, 1)):
            monkeypatch.setenv("AVIFGPU_CHUNK_MB", str(chunk_mb))
            gpu = rebind(n)
            got = harness.gpu_read(gpu, d, planes, mem="host")
            if ref is None:
                ref = got
                want = harness.oracle_read(d, planes)
                if d.depth == 32:
                    assert np.all(np.abs(got.astype(np.float64) - want) <= 1e-4 * np.abs(want) + 1e-9)
                else:
                    assert np.array_equal(got, want)
            assert np.array_equal(got.view(np.uint8), ref.view(np.uint8)), (kw["colorspace"], n)


def _shim_save(gpu, d, src, max_data, output, chroma, matrix, primaries, downsampling=0, fail_at_row=None):
    host = FakeHost(d.width, d.height, d.depth, d.planes, max_data=max_data, image=src, fail_at_row=fail_at_row)
    opts = H.SaveUIOptions(imageBitDepth=d.bit_depth, hdrTransferFunction=d.transfer, pq=H.PQOptions(d.peak_nits),
                           chromaSubsampling=chroma, lossless=0, chromaDownsampling=downsampling)
    img = H.Image()
    code = gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host.fr), d.alpha_state, ctypes.byref(opts), output, matrix,
                                                  primaries, ctypes.byref(img))
    return host, img, code


def _img_planes(img, d):
    out = {}
    ssz = 2 if d.bit_depth > 8 else 1
    for pl, (w, xs, ys) in harness.write_planes(d).items():
        h = (d.height + ys) >> ys
        raw = (ctypes.c_uint8 * (img.stride[pl] * h)).from_address(img.plane[pl])
        a = np.frombuffer(raw, dtype=np.uint8).reshape(h, img.stride[pl])[:, :w * ssz]
        out[pl] = a.view(np.uint16).copy() if ssz == 2 else a.copy()
    return out


@pytest.mark.parametrize("name,downsampling", [("c4-444", 0), ("c5-420-near", 0), ("c2-420", 1), ("rgba16-ref", 0)])
def test_shim_save_n_contexts_equal_one(rebind, name, downsampling):
    """Through the FormatRecord shim: tiles dealt round-robin to 1, 2, 4, 8 contexts -> the same heif_image planes, which are
    also the oracle's (bit-exact for integer documents)."""
    kw = dict(WRITE[name])
    if kw.get("output") == pkg.OUT_YCBCR and kw.get("chroma") != pkg.CHROMA_444:
        kw["chroma_downsampling"] = pkg.DOWNSAMPLE_AVERAGE if downsampling else pkg.DOWNSAMPLE_NEAREST
    d = pkg.WriteDesc(**kw)
    src = harness.make_write_source(d, seed=21)
    want = harness.oracle_write(d, src)
    row_bytes = d.width * d.planes * d.depth // 8
    ref = None
    for n in (1, 2, 4, 8):
        gpu = rebind(n)
        for max_data in (row_bytes * 6, row_bytes * 37):
            host, img, code = _shim_save(gpu, d, src, max_data, d.output, d.chroma, d.matrix_coefficients, d.color_primaries,
                                         downsampling)
            assert code == 0, gpu.lib.avifgpu_last_error()
            got = _img_planes(img, d)
            gpu.lib.avifgpu_image_free(ctypes.byref(img))
            tops = [r[0] for r in host.rects]
            assert tops == sorted(tops) and host.rects[0][0] == 0 and host.rects[-1][2] == d.height
            assert all(a[2] == b[0] for a, b in zip(host.rects[:-1], host.rects[1:]))
            if ref is None:
                ref = got
                if d.depth != 32:
                    for pl in want:
                        assert np.array_equal(got[pl], want[pl]), (name, pl)
                else:
                    st = harness.compare_write(d, want, got)
                    assert st["max_abs"] <= 1 and harness.t2_exact_ok(st, harness.T2_MIN_EXACT[d.bit_depth]), st
            for pl in ref:
                assert np.array_equal(got[pl], ref[pl]), (name, n, max_data, pl)


def test_shim_open_n_contexts_equal_one(rebind):
    d = pkg.ReadDesc(width=264, height=151, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_420, bit_depth=12, depth=16,
                     alpha_state=pkg.ALPHA_STRAIGHT, matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
    planes = harness.make_read_source(d, seed=8)
    want = harness.oracle_read(d, planes)
    for n in (1, 2, 4, 8):
        gpu = rebind(n)
        host = FakeHost(d.width, d.height, d.depth, 4, max_data=264 * 8 * 10)
        img = H.Image(width=d.width, height=d.height, colorspace=d.colorspace, chroma=d.chroma, bit_depth=d.bit_depth)
        for pl, a in planes.items():
            img.plane[pl] = a.ctypes.data
            img.stride[pl] = a.strides[0]
        nclx = H.Nclx(d.color_primaries, d.transfer_characteristics, d.matrix_coefficients, d.full_range_flag)
        code = gpu.lib.avifgpu_host_read_heif_image(ctypes.byref(img), d.alpha_state, ctypes.byref(nclx), None, ctypes.byref(host.fr))
        assert code == 0, gpu.lib.avifgpu_last_error()
        assert np.array_equal(host.image, want), n
        assert [r[0] for r in host.rects] == sorted(r[0] for r in host.rects)     # the host sees the tiles in row order


def test_shim_error_drains_all_contexts(rebind):
    """A host error in the middle of a multi-context save comes back unchanged and leaves no tile in flight: the next
    conversion on the same contexts is correct."""
    gpu = rebind(4)
    d = pkg.WriteDesc(**WRITE["c2-420"])
    src = harness.make_write_source(d, seed=2)
    row_bytes = d.width * 3
    host, img, code = _shim_save(gpu, d, src, row_bytes * 8, d.output, d.chroma, d.matrix_coefficients, d.color_primaries, 1,
                                 fail_at_row=100)
    assert code == -36
    gpu.lib.avifgpu_image_free(ctypes.byref(img))
    host, img, code = _shim_save(gpu, d, src, row_bytes * 8, d.output, d.chroma, d.matrix_coefficients, d.color_primaries, 1)
    assert code == 0
    want = harness.oracle_write(d, src)
    got = _img_planes(img, d)
    gpu.lib.avifgpu_image_free(ctypes.byref(img))
    for pl in want:
        assert np.array_equal(got[pl], want[pl])


def test_pinned_and_pageable_buffers_agree(rebind):
    """Page-locked caller memory is the DMA source / target itself, pageable memory is bounced by the worker: same bytes."""
    import torch
    gpu = rebind(2)
    d = pkg.WriteDesc(**WRITE["c4-444"])
    src = harness.make_write_source(d, seed=4)
    pageable = harness.gpu_write(gpu, d, src, mem="host")
    p_src = torch.from_numpy(src).pin_memory()
    outs = {pl: torch.zeros(((d.height + ys) >> ys, w * 2), dtype=torch.uint8).pin_memory()
            for pl, (w, xs, ys) in harness.write_planes(d).items()}
    gpu.write_rows(d, 0, d.height, p_src.data_ptr(), p_src.stride(0) * 4, [outs[i].data_ptr() if i in outs else None for i in range(4)],
                   [outs[i].stride(0) if i in outs else 0 for i in range(4)], mem=pkg.MEM_HOST)
    for pl in pageable:
        assert np.array_equal(outs[pl].numpy().view(np.uint16), pageable[pl]), pl


def test_rebind_and_device_pointers_after_multi(rebind):
    """Device-pointer launches keep working (on the caller's stream) while several contexts are bound, and after re-binding."""
    gpu = rebind(3)
    d = pkg.WriteDesc(**WRITE["rgba16-ref"])
    src = harness.make_write_source(d)
    a = harness.gpu_write(gpu, d, src, mem="device")
    gpu = rebind(1)
    b = harness.gpu_write(gpu, d, src, mem="device")
    want = harness.oracle_write(d, src)
    assert np.array_equal(a[0], want[0]) and np.array_equal(b[0], want[0])
    bad = (ctypes.c_int32 * 1)(99)
    assert gpu.lib.avifgpu_init_devices(bad, 1) == pkg.formatBadParameters
    assert gpu.lib.avifgpu_device_count() == 1          # a rejected list leaves the current binding alone


def test_fullsize_c4_eight_contexts_equal_one(rebind):
    """BASELINE.json configs[3] at full size (8192 x 8192 RGB f32 -> 10-bit PQ YCbCr 4:4:4: 805 MB in, 403 MB out) from host memory:
    the planes produced by 8 contexts (the 8-GPU split of one image; here 8 x 2 workers on the visible device(s)) are
    byte-identical to the 1-context planes, and every sample of them matches the oracle (T2 bar at 10 bit)."""
    import torch
    W = H_ = 8192
    d = pkg.WriteDesc(width=W, height=H_, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                      alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                      matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
    g = torch.Generator().manual_seed(1234)
    src = torch.rand((H_, W * 3), generator=g, dtype=torch.float32).pin_memory()
    outs = {}
    for n in (1, 8):
        gpu = rebind(n)
        planes = [torch.zeros((H_, W * 2), dtype=torch.uint8).pin_memory() for _ in range(3)]
        gpu.write_rows(d, 0, H_, src.data_ptr(), src.stride(0) * 4, [p.data_ptr() for p in planes] + [None],
                       [p.stride(0) for p in planes] + [0], mem=pkg.MEM_HOST)
        outs[n] = planes
    for a, b in zip(outs[1], outs[8]):
        assert torch.equal(a, b)
    # ... and the WHOLE frame against the oracle on every host core (round 6; rounds 3-5 compared a 12-row stripe across one cut)
    from test_gpu_fullsize import _oracle_frame
    want = _oracle_frame(d, src.numpy())
    for pl in range(3):
        got = outs[8][pl].numpy().view(np.uint16)
        bad, worst = 0, 0
        for r in range(0, H_, 1024):
            diff = np.abs(got[r:r + 1024].astype(np.int32) - want[pl][r:r + 1024, :W].astype(np.int32))
            bad += int(np.count_nonzero(diff))
            worst = max(worst, int(diff.max()))
        exact = 1.0 - bad / (H_ * W)
        assert worst <= 1 and exact >= 0.9995, (pl, worst, exact)
