"""The plug-in's own call sequence at BASELINE size: an 8192 x 8192 document saved and opened THROUGH THE FormatRecord PROTOCOL
(include/avifgpu_host.h, csrc/host_shim.cpp: multi-row theRect32 tiles sized from maxData, advanceState() per tile, the bound contexts'
staging slots and DMA under it) and the WHOLE result compared with the oracle run on every host core.  tests/test_gpu_host_shim.py
proves the protocol on 67 x 45 documents and tests/test_gpu_fullsize*.py prove the kernels on device-resident frames; the host-memory
pipeline under the shim (sub-tiles dealt over workers and slots, planes gathered in place, hundreds of DMA transfers per save) meets a
whole frame only here.  Reference loops reproduced: WriteHeifImage.cpp:1017-1135 (save), ReadHeifImage.cpp:83-400 + YuvDecode.cpp (open).

  integer documents: every byte equal;  32-bit saves: max |dcode| <= 1, exact >= 99.9 % (12 bit) per plane;  32-bit opens: the T2 read bar."""
import ctypes

import numpy as np
import pytest

import harness
from fake_host import FakeHost
from test_gpu_fullsize import _oracle_frame as _oracle_write_frame

pkg = harness.pkg
H = pkg.host
pytestmark = pytest.mark.gpu
W = HT = 8192
MAX_DATA = 64 << 20             # the host's maxData: 682-row tiles of an RGB f32 document (13 per save), 2730-row tiles of an RGB8 one


def _source(depth, planes, seed=1234):
    rng = np.random.default_rng(seed)
    if depth == 8:
        return rng.integers(0, 256, size=(HT, W * planes), dtype=np.uint8)
    src = rng.random((HT, W * planes), dtype=np.float32)
    m = rng.random((HT, W * planes), dtype=np.float32)
    np.putmask(src, m < 0.10, 1.0 + 11.5 * src)               # 10 % highlights up to 12.5 ...
    np.putmask(src, m > 0.999, -0.01 * src)                   # ... 0.1 % small negatives
    return src


SAVES = {
    "RGB-f32-12bit-PQ-reference-handoff (what integration/ ships by default)": dict(depth=32, bits=12, output=pkg.OUT_REFERENCE, chroma=pkg.CHROMA_422,
                                                                                    transfer=pkg.TRANSFER_PQ, matrix=pkg.MATRIX_BT2020_NCL),
    "RGB-f32-12bit-PQ-422-planes (the fused default HDR save)": dict(depth=32, bits=12, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                                                      transfer=pkg.TRANSFER_PQ, matrix=pkg.MATRIX_BT2020_NCL),
    "RGB8-8bit-422-planes-601 (the fused default SDR save)": dict(depth=8, bits=8, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422,
                                                                   transfer=pkg.TRANSFER_CLIP, matrix=pkg.MATRIX_BT601),
}


@pytest.mark.parametrize("name", list(SAVES))
def test_fullsize_save_through_the_format_record_protocol(gpu, name):
    c = SAVES[name]
    hdr = c["depth"] == 32
    d = pkg.WriteDesc(width=W, height=HT, depth=c["depth"], planes=3, bit_depth=c["bits"], transfer=c["transfer"], peak_nits=80,
                      alpha_state=pkg.ALPHA_NONE, output=c["output"], chroma=c["chroma"], chroma_downsampling=pkg.DOWNSAMPLE_NEAREST,
                      matrix_coefficients=c["matrix"], color_primaries=pkg.PRIMARIES_BT2020 if hdr else pkg.PRIMARIES_BT709)
    src = _source(c["depth"], 3)
    host = FakeHost(W, HT, c["depth"], 3, max_data=MAX_DATA, image=src)
    opts = H.SaveUIOptions(imageBitDepth=c["bits"], hdrTransferFunction=c["transfer"], pq=H.PQOptions(80), chromaSubsampling=c["chroma"], lossless=0)
    img = H.Image()
    code = gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host.fr), pkg.ALPHA_NONE, ctypes.byref(opts), c["output"], c["matrix"],
                                                  pkg.PRIMARIES_BT2020 if hdr else pkg.PRIMARIES_BT709, ctypes.byref(img))
    assert code == 0, gpu.lib.avifgpu_last_error()
    try:
        # the protocol at this size: ascending full-width tiles that cover [0, H), each within maxData
        assert host.rects[0][0] == 0 and host.rects[-1][2] == HT and len(host.rects) > 1
        assert all(a[2] == b[0] for a, b in zip(host.rects[:-1], host.rects[1:]))
        assert all(r[1] == 0 and r[3] == W and (r[2] - r[0]) * host.fr.rowBytes <= MAX_DATA for r in host.rects)
        want = _oracle_write_frame(d, src)
        ssz = 2 if d.bit_depth > 8 else 1
        for pl, (w, xs, ys) in harness.write_planes(d).items():
            h = (HT + ys) >> ys
            raw = (ctypes.c_uint8 * (img.stride[pl] * h)).from_address(img.plane[pl])
            got = np.frombuffer(raw, dtype=np.uint8).reshape(h, img.stride[pl])[:, :w * ssz]
            got = got.view(np.uint16) if ssz == 2 else got
            wnt = want[pl][:h, :w]
            if not hdr:
                assert np.array_equal(got, wnt), (name, pl)
                continue
            bad, worst = 0, 0
            for r in range(0, h, 1024):                        # 1024 rows of int32 temporaries at a time
                diff = np.abs(got[r:r + 1024].astype(np.int32) - wnt[r:r + 1024].astype(np.int32))
                bad += int(np.count_nonzero(diff))
                worst = max(worst, int(diff.max()))
            exact = 1.0 - bad / (h * w)
            print(f"{name} plane {pl}: {h}x{w} samples in {len(host.rects)} tiles, exact {exact:.6f}, max |dcode| {worst}")
            assert worst <= 1 and exact >= 0.999, (name, pl, worst, exact)
    finally:
        gpu.lib.avifgpu_image_free(ctypes.byref(img))


OPENS = {
    "12bit-422-PQ -> RGB f32 (what the default HDR save decodes to)": dict(colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_422, bit_depth=12, depth=32,
                                                                           matrix_coefficients=9, color_primaries=9, transfer_characteristics=16, pq_peak_nits=80),
    "8bit-420-709 -> RGB8": dict(colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_420, bit_depth=8, depth=8, matrix_coefficients=1),
}


@pytest.mark.parametrize("name", list(OPENS))
def test_fullsize_open_through_the_format_record_protocol(gpu, name):
    import oracle_binding
    d = pkg.ReadDesc(width=W, height=HT, alpha_state=pkg.ALPHA_NONE, **OPENS[name])
    rng = np.random.default_rng(4321)
    ssz = 2 if d.bit_depth > 8 else 1
    planes = {}
    for pl, (w, xs, ys) in harness.read_planes(d).items():
        planes[pl] = rng.integers(0, 1 << d.bit_depth, size=((HT + ys) >> ys, w), dtype=np.uint16 if ssz == 2 else np.uint8)
    nch = harness.read_channels(d)
    host = FakeHost(W, HT, d.depth, nch, max_data=MAX_DATA)
    img = H.Image(width=W, height=HT, colorspace=d.colorspace, chroma=d.chroma, bit_depth=d.bit_depth)
    for pl, a in planes.items():
        img.plane[pl], img.stride[pl] = a.ctypes.data, a.strides[0]
    nclx = H.Nclx(d.color_primaries, d.transfer_characteristics, d.matrix_coefficients, d.full_range_flag)
    load = H.LoadUIOptions(hlg=H.HLGOptions(0, 1.2, 1000), pq=H.PQOptions(80)) if d.depth == 32 else None
    code = gpu.lib.avifgpu_host_read_heif_image(ctypes.byref(img), pkg.ALPHA_NONE, ctypes.byref(nclx), ctypes.byref(load) if load is not None else None,
                                                ctypes.byref(host.fr))
    assert code == 0, gpu.lib.avifgpu_last_error()
    assert host.rects[0][0] == 0 and host.rects[-1][2] == HT and len(host.rects) > 1
    assert all(a[2] == b[0] for a, b in zip(host.rects[:-1], host.rects[1:]))
    L = oracle_binding.load()
    ptrs, strides = [None] * 4, [0] * 4
    for pl, a in planes.items():
        ptrs[pl], strides[pl] = a.ctypes.data, a.strides[0]
    want = np.empty_like(host.image)
    n = ctypes.c_int32(0)
    assert L.oracle_read_image_all_cores(ctypes.byref(d), ctypes.byref(pkg.planes4(ptrs)), ctypes.byref(pkg.strides4(strides)),
                                         want.ctypes.data, want.strides[0], ctypes.byref(n)) == 0
    if d.depth != 32:
        assert np.array_equal(host.image, want), name
        return
    worst = 0.0
    for r in range(0, HT, 1024):
        g64, w64 = host.image[r:r + 1024].astype(np.float64), want[r:r + 1024].astype(np.float64)
        err = np.abs(g64 - w64)
        assert np.isfinite(g64).all() and np.all(err <= 1e-4 * np.abs(w64) + 1e-9), (name, r)
        worst = max(worst, float((err / np.maximum(np.abs(w64), 1e-6)).max()))
    print(f"{name}: {HT}x{W * nch} samples in {len(host.rects)} tiles, max relative error against the oracle {worst:.2e}")
