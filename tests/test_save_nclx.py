"""avifgpu_host_save_nclx = AddColorProfileToImage (WriteMetadata.cpp:107-149): the nclx the plug-in attaches, hence the matrix a
fused YCbCr hand-off must use.  Host logic only (no GPU)."""
import ctypes

import pytest

import harness

pkg = harness.pkg
H = pkg.host
MONO, RGB = H.plugInModeGrayScale, H.plugInModeRGBColor


def _ask(depth, mode, transfer, lossless):
    lib = pkg.load()
    fr = H.FormatRecord()
    fr.depth, fr.imageMode, fr.planes = depth, mode, (1 if mode == MONO else 3)
    o = H.SaveUIOptions(imageBitDepth=10, hdrTransferFunction=transfer, pq=H.PQOptions(1000), chromaSubsampling=pkg.CHROMA_420,
                        lossless=lossless, convertToRec2020=0, convertToSRGB=0)
    n = H.Nclx()
    rc = lib.avifgpu_host_save_nclx(ctypes.byref(fr), ctypes.byref(o), ctypes.byref(n))
    return rc, (n.color_primaries, n.transfer_characteristics, n.matrix_coefficients, n.full_range_flag)


SDR = (pkg.PRIMARIES_BT709, pkg.TC_SRGB, pkg.MATRIX_BT601, 1)


@pytest.mark.parametrize("depth,mode,transfer,lossless,expected", [
    (8, RGB, pkg.TRANSFER_CLIP, 0, SDR),
    (16, RGB, pkg.TRANSFER_PQ, 0, SDR),                                   # the transfer option only applies to 32-bit documents
    (32, H.plugInModeRGB96, pkg.TRANSFER_CLIP, 0, SDR),
    (32, H.plugInModeRGB96, pkg.TRANSFER_PQ, 0, (pkg.PRIMARIES_BT2020, pkg.TC_PQ, pkg.MATRIX_BT2020_NCL, 1)),
    (32, H.plugInModeRGB96, pkg.TRANSFER_SMPTE428, 0, (pkg.PRIMARIES_BT2020, pkg.TC_SMPTE428, pkg.MATRIX_BT2020_NCL, 1)),
    (8, RGB, pkg.TRANSFER_CLIP, 1, (pkg.PRIMARIES_BT709, pkg.TC_SRGB, pkg.MATRIX_RGB_GBR, 1)),
    (32, H.plugInModeRGB96, pkg.TRANSFER_PQ, 1, (pkg.PRIMARIES_BT2020, pkg.TC_PQ, pkg.MATRIX_RGB_GBR, 1)),
    (8, MONO, pkg.TRANSFER_CLIP, 1, SDR),                                  # lossless monochrome keeps BT.601
])
def test_save_nclx_matches_the_plugin(depth, mode, transfer, lossless, expected):
    rc, got = _ask(depth, mode, transfer, lossless)
    assert rc == 0 and got == expected


def test_save_nclx_rejects_hlg_like_the_plugin():
    rc, _ = _ask(32, H.plugInModeRGB96, pkg.TRANSFER_HLG, 0)
    assert rc == pkg.writErr


def test_shim_rejects_hlg_saves_like_the_plugin():
    """Every 32-bit save loop of the reference throws "Unsupported color transfer function." for HLG
    (WriteHeifImage.cpp:1088-1089); the shim does so before any tile is requested (so this runs without a GPU)."""
    from fake_host import FakeHost
    import numpy as np
    host = FakeHost(8, 2, 32, 3, image=np.zeros((2, 24), dtype=np.float32))
    o = H.SaveUIOptions(imageBitDepth=10, hdrTransferFunction=pkg.TRANSFER_HLG, pq=H.PQOptions(1000), chromaSubsampling=pkg.CHROMA_420,
                        lossless=0, convertToRec2020=0, convertToSRGB=0)
    img = H.Image()
    rc = pkg.load().avifgpu_host_create_heif_image(ctypes.byref(host.fr), pkg.ALPHA_NONE, ctypes.byref(o), pkg.OUT_REFERENCE,
                                                   pkg.MATRIX_BT2020_NCL, pkg.PRIMARIES_BT2020, ctypes.byref(img))
    assert rc == pkg.writErr and host.rects == []
    assert b"Unsupported color transfer function" in pkg.load().avifgpu_last_error()
