"""The FormatRecord-protocol shim (include/avifgpu_host.h, csrc/host_shim.cpp) against a fake Photoshop host:
same callbacks and rectangles as the reference row loops, multi-row tiles, OSErr behaviour, results identical to the
oracle's whole-frame conversion."""
import ctypes

import numpy as np
import pytest

import harness
from fake_host import FakeHost

pkg = harness.pkg
H = pkg.host
pytestmark = pytest.mark.gpu


def _save(gpu, desc_kw, max_data, output, abort_after=None, fail_at_row=None, matrix=pkg.MATRIX_BT601, chroma=pkg.CHROMA_444):
    # the shim's stage B is libheif 1.14.0's: co-sited chroma sample (SaveUIOptions.chromaDownsampling = 0)
    d = pkg.WriteDesc(**dict(desc_kw, chroma_downsampling=pkg.DOWNSAMPLE_NEAREST))
    src = harness.make_write_source(d)
    host = FakeHost(d.width, d.height, d.depth, d.planes, max_data=max_data, image=src, abort_after=abort_after,
                    fail_at_row=fail_at_row)
    opts = H.SaveUIOptions(imageBitDepth=d.bit_depth, hdrTransferFunction=d.transfer, pq=H.PQOptions(d.peak_nits),
                           chromaSubsampling=chroma, lossless=0)
    img = H.Image()
    code = gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host.fr), d.alpha_state, ctypes.byref(opts), output, matrix,
                                                  pkg.PRIMARIES_BT709, ctypes.byref(img))
    return d, src, host, img, code


def _planes_of(img, desc):
    out = {}
    ssz = 2 if desc.bit_depth > 8 else 1
    for pl, (w, xs, ys) in harness.write_planes(desc).items():
        h = (desc.height + ys) >> ys
        raw = (ctypes.c_uint8 * (img.stride[pl] * h)).from_address(img.plane[pl])
        a = np.frombuffer(raw, dtype=np.uint8).reshape(h, img.stride[pl])[:, :w * ssz]
        out[pl] = a.view(np.uint16).copy() if ssz == 2 else a.copy()
    return out


@pytest.mark.parametrize("kw,output,chroma", [
    (dict(width=67, height=45, depth=8, planes=4, bit_depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED, output=pkg.OUT_REFERENCE), pkg.OUT_REFERENCE, pkg.CHROMA_444),
    (dict(width=67, height=45, depth=16, planes=2, bit_depth=12, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_REFERENCE), pkg.OUT_REFERENCE, pkg.CHROMA_444),
    (dict(width=67, height=45, depth=8, planes=3, bit_depth=8, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT601), pkg.OUT_YCBCR, pkg.CHROMA_420),
    (dict(width=67, height=45, depth=16, planes=4, bit_depth=10, alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_422, matrix_coefficients=pkg.MATRIX_BT601), pkg.OUT_YCBCR, pkg.CHROMA_422),
])
def test_save_matches_oracle_bit_exact(gpu, kw, output, chroma):
    for max_data in (0, 67 * 4 * 2 * 7, 1):          # whole image, 7-row tiles, degenerate (1 row / 2 rows for 4:2:0)
        d, src, host, img, code = _save(gpu, kw, max_data, output, chroma=chroma)
        assert code == 0, gpu.lib.avifgpu_last_error()
        want = harness.oracle_write(d, src)
        got = _planes_of(img, d)
        for pl in want:
            assert np.array_equal(got[pl], want[pl]), (max_data, pl)
        # protocol: rectangles tile [0, H) in order, full width, each within maxData, abortProc polled once per tile
        tops = [r[0] for r in host.rects]
        assert tops[0] == 0 and host.rects[-1][2] == d.height
        assert all(a[2] == b[0] for a, b in zip(host.rects[:-1], host.rects[1:]))
        assert all(r[1] == 0 and r[3] == d.width for r in host.rects)
        assert host.polls == len(host.rects)
        if max_data > 1:
            assert all((r[2] - r[0]) * host.fr.rowBytes <= max(max_data, 2 * host.fr.rowBytes) for r in host.rects)
        assert host.fr.loPlane == 0 and host.fr.hiPlane == d.planes - 1 and host.fr.rowBytes == d.width * d.planes * d.depth // 8
        assert img.premultiplied_alpha == (1 if d.alpha_state == pkg.ALPHA_PREMULTIPLIED else 0)
        gpu.lib.avifgpu_image_free(ctypes.byref(img))


def test_save_hdr_float_tier(gpu):
    kw = dict(width=264, height=96, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80,
              alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
              color_primaries=pkg.PRIMARIES_BT2020)
    d, src, host, img, code = _save(gpu, kw, 264 * 12 * 10, pkg.OUT_YCBCR, matrix=pkg.MATRIX_BT2020_NCL)
    assert code == 0
    st = harness.compare_write(d, harness.oracle_write(d, src), _planes_of(img, d))
    assert st["max_abs"] <= 1 and harness.t2_exact_ok(st, harness.T2_MIN_EXACT[d.bit_depth]), st
    gpu.lib.avifgpu_image_free(ctypes.byref(img))


def test_cancel_and_host_error(gpu):
    kw = dict(width=64, height=64, depth=8, planes=3, bit_depth=8, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
    d, src, host, img, code = _save(gpu, kw, 64 * 3 * 8, pkg.OUT_REFERENCE, abort_after=3)
    assert code == pkg.userCanceledErr and len(host.rects) == 3           # WriteHeifImage.cpp:1019-1022
    gpu.lib.avifgpu_image_free(ctypes.byref(img))
    d, src, host, img, code = _save(gpu, kw, 64 * 3 * 8, pkg.OUT_REFERENCE, fail_at_row=20)
    assert code == -36                                                    # OSErrException::ThrowIfError(advanceState())
    gpu.lib.avifgpu_image_free(ctypes.byref(img))
    # alpha state that disagrees with the document's planes
    host = FakeHost(8, 8, 8, 3, image=np.zeros((8, 24), np.uint8))
    opts = H.SaveUIOptions(imageBitDepth=8, hdrTransferFunction=pkg.TRANSFER_CLIP, pq=H.PQOptions(80), chromaSubsampling=3)
    img = H.Image()
    assert gpu.lib.avifgpu_host_create_heif_image(ctypes.byref(host.fr), pkg.ALPHA_STRAIGHT, ctypes.byref(opts), 0, 6, 1,
                                                  ctypes.byref(img)) == pkg.formatBadParameters


@pytest.mark.parametrize("kw", [
    dict(width=67, height=45, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_420, bit_depth=8, depth=8, alpha_state=pkg.ALPHA_PREMULTIPLIED, matrix_coefficients=pkg.MATRIX_BT709),
    dict(width=67, height=45, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_422, bit_depth=10, depth=16, alpha_state=pkg.ALPHA_NONE, matrix_coefficients=pkg.MATRIX_BT2020_NCL, full_range_flag=0),
    dict(width=67, height=45, colorspace=pkg.COLORSPACE_MONOCHROME, chroma=pkg.CHROMA_MONOCHROME, bit_depth=12, depth=16, alpha_state=pkg.ALPHA_STRAIGHT),
    dict(width=67, height=45, colorspace=pkg.COLORSPACE_RGB, chroma=pkg.CHROMA_444, bit_depth=10, depth=16, alpha_state=pkg.ALPHA_PREMULTIPLIED, matrix_coefficients=pkg.MATRIX_RGB_GBR),
])
def test_open_matches_oracle_bit_exact(gpu, kw):
    d = pkg.ReadDesc(**kw)
    planes = harness.make_read_source(d)
    want = harness.oracle_read(d, planes)
    nch = harness.read_channels(d)
    for max_data in (0, 67 * nch * (d.depth // 8) * 6):
        host = FakeHost(d.width, d.height, d.depth, nch, max_data=max_data)
        img = H.Image(width=d.width, height=d.height, colorspace=d.colorspace, chroma=d.chroma, bit_depth=d.bit_depth)
        for pl, a in planes.items():
            img.plane[pl] = a.ctypes.data
            img.stride[pl] = a.strides[0]
        nclx = H.Nclx(d.color_primaries, d.transfer_characteristics, d.matrix_coefficients, d.full_range_flag)
        code = gpu.lib.avifgpu_host_read_heif_image(ctypes.byref(img), d.alpha_state, ctypes.byref(nclx), None, ctypes.byref(host.fr))
        assert code == 0, gpu.lib.avifgpu_last_error()
        assert np.array_equal(host.image, want), max_data
        assert host.rects[0][0] == 0 and host.rects[-1][2] == d.height
        if d.depth == 16:
            assert host.fr.maxValue == (32768 if d.colorspace != pkg.COLORSPACE_RGB else (1 << d.bit_depth) - 1)


def test_open_hdr_requires_nclx(gpu):
    host = FakeHost(16, 16, 32, 3)
    img = H.Image(width=16, height=16, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_444, bit_depth=10)
    planes = [np.zeros((16, 16), np.uint16) for _ in range(3)]
    for pl, a in enumerate(planes):
        img.plane[pl] = a.ctypes.data
        img.stride[pl] = a.strides[0]
    code = gpu.lib.avifgpu_host_read_heif_image(ctypes.byref(img), pkg.ALPHA_NONE, None, None, ctypes.byref(host.fr))
    assert code == pkg.readErr and b"nclxProfile is null" in gpu.lib.avifgpu_last_error()



@pytest.mark.parametrize("tc,load", [
    (pkg.TC_PQ, dict(pq=1000)), (pkg.TC_HLG, dict(ootf=1, gamma=1.4, peak=600)), (pkg.TC_HLG, dict(ootf=0)), (pkg.TC_SMPTE428, dict())],
    ids=["pq-1000nits", "hlg-ootf-gamma1.4-600nits", "hlg-no-ootf", "smpte428"])
def test_open_hdr_takes_the_load_options(gpu, tc, load):
    """A 32-bit open through the shim: LoadUIOptions (Read.cpp:592-625 hands them to ReadHeifImage*ThirtyTwoBit) must reach the kernel --
    PQ nominal peak, HLG OOTF on/off with its display gamma and peak (ColorTransfer.cpp:192-205) -- tiles of 5 rows, against the oracle
    driven with the same values in its descriptor (T2 read bar)."""
    w, h = 131, 37
    opts = H.LoadUIOptions(hlg=H.HLGOptions(load.get("ootf", 0), load.get("gamma", 1.2), load.get("peak", 1000)), pq=H.PQOptions(load.get("pq", 80)))
    d = pkg.ReadDesc(width=w, height=h, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_420, bit_depth=10, depth=32, alpha_state=pkg.ALPHA_NONE,
                     matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020, transfer_characteristics=tc,
                     pq_peak_nits=opts.pq.nominalPeakBrightness, hlg_apply_ootf=opts.hlg.applyOOTF, hlg_display_gamma=opts.hlg.displayGamma,
                     hlg_peak_nits=opts.hlg.nominalPeakBrightness)
    planes = harness.make_read_source(d)
    want = harness.oracle_read(d, planes)
    host = FakeHost(w, h, 32, 3, max_data=w * 3 * 4 * 5)
    img = H.Image(width=w, height=h, colorspace=d.colorspace, chroma=d.chroma, bit_depth=d.bit_depth)
    for pl, a in planes.items():
        img.plane[pl], img.stride[pl] = a.ctypes.data, a.strides[0]
    nclx = H.Nclx(d.color_primaries, d.transfer_characteristics, d.matrix_coefficients, d.full_range_flag)
    code = gpu.lib.avifgpu_host_read_heif_image(ctypes.byref(img), pkg.ALPHA_NONE, ctypes.byref(nclx), ctypes.byref(opts), ctypes.byref(host.fr))
    assert code == 0, gpu.lib.avifgpu_last_error()
    assert len(host.rects) > 3
    got = host.image.astype(np.float64)
    assert np.all(np.abs(got - want) <= 1e-4 * np.abs(want) + 1e-9), (tc, load, float(np.abs(got - want).max()))
    # the options matter: the same planes decoded with the defaults (80 nits; gamma 1.2 / 1000 nits) differ wherever they are used
    if load.get("pq") or load.get("ootf"):
        dflt = pkg.ReadDesc(width=w, height=h, colorspace=d.colorspace, chroma=d.chroma, bit_depth=10, depth=32, alpha_state=pkg.ALPHA_NONE,
                            matrix_coefficients=d.matrix_coefficients, color_primaries=d.color_primaries, transfer_characteristics=tc)
        assert not np.allclose(harness.oracle_read(dflt, planes), want)
