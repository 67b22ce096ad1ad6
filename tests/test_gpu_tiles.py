"""Row-tile sharding on the GPU: N even-row tiles (what N ranks / N GPUs each convert) reproduce the 1-GPU frame
byte for byte, for 1/2/4/8 tiles, both directions.  Tiles are launched on separate HIP streams of one device when
fewer than N GPUs exist (multi-GPU row)."""
import numpy as np
import pytest

import harness

pkg = harness.pkg
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("chroma", [pkg.CHROMA_444, pkg.CHROMA_420])
def test_write_tiles_equal_whole(gpu, world, chroma):
    H, W = 150, 264            # 150/8 = 18.75: uneven tiles with even starts, last tile odd height
    d = pkg.WriteDesc(width=W, height=H, depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=1000,
                      alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR, chroma=chroma,
                      matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
    src = harness.make_write_source(d)
    whole = harness.gpu_write(gpu, d, src)
    tiles = pkg.sharding.all_tiles(H, world, even=True)
    assert tiles[0][0] == 0 and sum(n for _, n in tiles) == H and all(r0 % 2 == 0 for r0, _ in tiles)
    for pl, (w, xs, ys) in harness.write_planes(d).items():
        parts = [harness.gpu_write(gpu, d, src, row0=r0, nrows=n)[pl] for r0, n in tiles if n]
        assert np.array_equal(np.concatenate(parts, axis=0), whole[pl]), (world, pl)


@pytest.mark.parametrize("world", [2, 8])
def test_read_tiles_equal_whole(gpu, world):
    H, W = 150, 264
    d = pkg.ReadDesc(width=W, height=H, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_420, bit_depth=10, depth=16,
                     alpha_state=pkg.ALPHA_PREMULTIPLIED, matrix_coefficients=pkg.MATRIX_BT709)
    planes = harness.make_read_source(d)
    whole = harness.gpu_read(gpu, d, planes)
    parts = [harness.gpu_read(gpu, d, planes, row0=r0, nrows=n) for r0, n in pkg.sharding.all_tiles(H, world) if n]
    assert np.array_equal(np.concatenate(parts, axis=0), whole)


def test_concurrent_streams(gpu):
    """8 tiles in flight on 8 HIP streams of one device == sequential result (no shared mutable state in the library)."""
    import torch
    H, W = 512, 1024
    d = pkg.WriteDesc(width=W, height=H, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                      alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                      matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
    src = harness.make_write_source(d)
    whole = harness.gpu_write(gpu, d, src)
    dev = f"cuda:{gpu.device}"
    d_src = torch.from_numpy(src).to(dev)
    outs = [torch.zeros((H, W * 2), dtype=torch.uint8, device=dev) for _ in range(3)]
    torch.cuda.synchronize(dev)
    streams = [torch.cuda.Stream(dev) for _ in range(8)]
    for (r0, n), st in zip(pkg.sharding.all_tiles(H, 8), streams):
        gpu.write_rows(d, r0, n, d_src[r0].data_ptr(), d_src.stride(0) * 4, [o[r0].data_ptr() for o in outs] + [None],
                       [o.stride(0) for o in outs] + [0], mem=pkg.MEM_DEVICE, stream=st.cuda_stream)
    torch.cuda.synchronize(dev)
    for pl in range(3):
        assert np.array_equal(outs[pl].cpu().numpy().view(np.uint16), whole[pl])
