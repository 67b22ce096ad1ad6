"""The packed 8-bit fast paths (write_px FAST8: RGB8 / RGBA8 -> u8 Y,Cb,Cr(,A) planes; read_px PACKED8: u8 YCbCr(A) -> RGB8 / RGBA8)
only exist in the ALIGNED instantiations (16-byte aligned pointers and strides), which the small odd-width cases of
tests/cases.py never reach with their tight rows.  Here the rows are PADDED to 16 bytes the way libheif pads its planes, so odd
and ragged widths run the aligned kernels: the one ragged lane of a row (replicated last pixel, byte-wise stores), odd heights
(replicated last row), both chroma down-sampling modes, premultiplied alpha.  Bit-exact against the oracle."""
import numpy as np
import pytest

import harness

pkg = harness.pkg
pytestmark = pytest.mark.gpu


def align(v, a):
    return (v + a - 1) // a * a


def gpu_write_padded(gpu, desc, src):
    """Like harness.gpu_write(mem='device'), but the source rows sit at a 16-byte padded stride on the device."""
    import torch
    dev = f"cuda:{gpu.device}"
    H, rowb = src.shape[0], src.shape[1] * src.itemsize
    stride = align(rowb, 16)
    padded = np.full((H, stride), 0x5A, dtype=np.uint8)
    padded[:, :rowb] = src.view(np.uint8).reshape(H, rowb)
    d_src = torch.from_numpy(padded.reshape(-1)).to(dev)
    bufs = harness._alloc_write_out(desc, H)
    d_out = {pl: torch.from_numpy(b.view(np.uint8).reshape(-1).copy()).to(dev) for pl, b in bufs.items()}
    ptrs = [d_out[i].data_ptr() if i in d_out else None for i in range(4)]
    strides = [bufs[i].strides[0] if i in bufs else 0 for i in range(4)]
    gpu.write_rows(desc, 0, H, d_src.data_ptr(), stride, ptrs, strides, mem=pkg.MEM_DEVICE, stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    raw = {pl: d_out[pl].cpu().numpy().view(bufs[pl].dtype).reshape(bufs[pl].shape) for pl in bufs}
    return harness._trim(desc, raw, H, harness.write_planes), raw


WIDTHS = [1040, 1001, 1000, 24, 17, 2050, 2048, 8, 1032]
CHROMAS = [pkg.CHROMA_444, pkg.CHROMA_422, pkg.CHROMA_420]


@pytest.mark.parametrize("planes,alpha", [(3, pkg.ALPHA_NONE), (4, pkg.ALPHA_STRAIGHT), (4, pkg.ALPHA_PREMULTIPLIED)])
@pytest.mark.parametrize("chroma", CHROMAS)
@pytest.mark.parametrize("width", WIDTHS)
def test_write_u8_fast_path_padded_rows(gpu, planes, alpha, chroma, width):
    for height, near, matrix in ((7, False, pkg.MATRIX_BT601), (4, True, pkg.MATRIX_BT709)):
        kw = dict(width=width, height=height, depth=8, planes=planes, bit_depth=8, alpha_state=alpha, output=pkg.OUT_YCBCR,
                  chroma=chroma, matrix_coefficients=matrix, color_primaries=pkg.PRIMARIES_BT709)
        if near:
            kw["chroma_downsampling"] = pkg.DOWNSAMPLE_NEAREST
        d = pkg.WriteDesc(**kw)
        src = harness.make_write_source(d, seed=width + height)
        want = harness.oracle_write(d, src)
        # tuning word 7 = the library's choice: RGB8 rows of whole 8-pixel groups take the streaming kernel (round 5), everything else the
        # packed path of the generic kernel; tuning word 0 = no streaming kernels: the packed path on every case
        for variant in (7, 0):
            gpu.lib.avifgpu_set_hot_variant(variant)
            try:
                got, raw = gpu_write_padded(gpu, d, src)
                k = gpu.last_kernel()
            finally:
                gpu.lib.avifgpu_set_hot_variant(7)
            if variant == 7 and width % 8 == 0:
                assert ("write_rgb8_ycbcr_hot" if planes == 3 else "write_rgba8_ycbcra_hot") in k, k
            else:
                assert "aligned=1" in k and "depth=8" in k, k
            for pl in want:
                assert np.array_equal(want[pl], got[pl]), (k, pl, width, height, near, int(np.abs(want[pl].astype(int) - got[pl].astype(int)).max()))
            # nothing written beyond the valid samples of a row (the guard pattern of the padding survives)
            for pl, (w, xs, ys) in harness.write_planes(d).items():
                assert np.all(raw[pl][:, w:] == 0xA5), (k, pl, width)


@pytest.mark.parametrize("planes,alpha", [(4, pkg.ALPHA_STRAIGHT), (4, pkg.ALPHA_PREMULTIPLIED), (3, pkg.ALPHA_NONE)])
@pytest.mark.parametrize("chroma", CHROMAS)
@pytest.mark.parametrize("width", [1040, 1001, 24, 17, 2050, 1000, 8])
def test_write_16bit_document_to_u8_planes_padded_rows(gpu, planes, alpha, chroma, width):
    """A 16-bit document saved at 8 bit: rows of whole 8-pixel groups take the RGB16 / RGBA16 streaming kernels with u8 planes (rounds 5 / 6: the
    RGBA16 ones rescale and premultiply in integer arithmetic, rgba16_pixel_to8), everything else the generic kernel's packed footprint -- same
    cases either way, incl. samples beyond 32768."""
    for height, near in ((7, False), (4, True)):
        kw = dict(width=width, height=height, depth=16, planes=planes, bit_depth=8, alpha_state=alpha, output=pkg.OUT_YCBCR,
                  chroma=chroma, matrix_coefficients=pkg.MATRIX_BT601, color_primaries=pkg.PRIMARIES_BT709)
        if near:
            kw["chroma_downsampling"] = pkg.DOWNSAMPLE_NEAREST
        d = pkg.WriteDesc(**kw)
        src = harness.make_write_source(d, seed=width + height + planes)
        src.reshape(-1)[5::97] = 40000                     # beyond Photoshop's range: clamped like the reference's table index
        want = harness.oracle_write(d, src)
        # tuning word 7: RGB16 rows of whole 8-pixel groups take the RGB16 streaming kernels with u8 planes (round 5); word 0: the generic kernel
        for variant in (7, 0):
            gpu.lib.avifgpu_set_hot_variant(variant)
            try:
                got, raw = gpu_write_padded(gpu, d, src)
                k = gpu.last_kernel()
            finally:
                gpu.lib.avifgpu_set_hot_variant(7)
            if variant == 7 and planes == 3 and width % 8 == 0:
                assert "write_rgb16_ycbcr" in k and "to8" in k, k
            elif variant == 7 and planes == 4 and width % 8 == 0:           # round 6: RGBA16 -> u8 planes on the RGBA16 streaming kernels too
                assert "write_rgba16_ycbcra" in k and "to8" in k, k
            else:
                assert "aligned=1" in k and "depth=16" in k, k
            for pl in want:
                assert np.array_equal(want[pl], got[pl]), (k, pl, width, height, near, int(np.abs(want[pl].astype(int) - got[pl].astype(int)).max()))
            for pl, (w, xs, ys) in harness.write_planes(d).items():
                assert np.all(raw[pl][:, w:] == 0xA5), (k, pl, width)


@pytest.mark.parametrize("alpha", [pkg.ALPHA_NONE, pkg.ALPHA_STRAIGHT, pkg.ALPHA_PREMULTIPLIED])
@pytest.mark.parametrize("chroma", CHROMAS)
@pytest.mark.parametrize("width", [1040, 1001, 24, 2050])
def test_read_u8_fast_path_padded_rows(gpu, alpha, chroma, width):
    for height, full_range, matrix in ((7, True, pkg.MATRIX_BT601), (4, False, pkg.MATRIX_BT709)):
        d = pkg.ReadDesc(width=width, height=height, colorspace=pkg.COLORSPACE_YCBCR, chroma=chroma, bit_depth=8, depth=8,
                         alpha_state=alpha, matrix_coefficients=matrix, color_primaries=pkg.PRIMARIES_BT709,
                         transfer_characteristics=pkg.TC_SRGB, full_range_flag=1 if full_range else 0)
        planes = harness.make_read_source(d, seed=width + height)
        # 16-byte strides (libheif's own padding): re-pad every plane
        fixed = {}
        for pl, arr in planes.items():
            w = arr.shape[1]
            out = np.zeros((arr.shape[0], align(w, 16)), dtype=arr.dtype)
            out[:, :w] = arr
            fixed[pl] = out
        want = harness.oracle_read(d, fixed)
        got = harness.gpu_read(gpu, d, fixed, mem="device")
        assert "aligned=1" in gpu.last_kernel() and "depth=8" in gpu.last_kernel(), gpu.last_kernel()
        assert np.array_equal(want, got), (width, height, full_range)
