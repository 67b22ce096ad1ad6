"""ICC stage against COMMITTED vectors generated with the real Little CMS 2 (tests/golden/make_icc_vectors.py): the same
bit-exact / tier-2 bars as the live-library tests, but runnable where lcms2 is not installed.
CPU: host-built tables + numpy restatements of lcms2's integer evaluators.  GPU: the fused kernels."""
import ctypes
import os

import numpy as np
import pytest

import harness
from test_icc16 import _tetrahedral

pkg = harness.pkg
V = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icc_vectors.npz"))
NAMES = sorted({k.split(".")[0] for k in V.files})


def _icc(name):
    return V[name + ".icc"].tobytes()


@pytest.mark.parametrize("name", NAMES)
def test_8bit_tables_match_vectors(name):
    icc = _icc(name)
    sh = pkg.IccShaper8()
    assert pkg.load().avifgpu_icc_prepare_shaper8(icc, len(icc), ctypes.byref(sh)) == 0
    s1, M, s2 = np.array(sh.shaper1, dtype=np.int64), np.array(sh.matrix, dtype=np.int64), np.array(sh.shaper2, dtype=np.uint8)
    px = V[name + ".in8"].reshape(-1, 3)
    R, G, B = s1[0][px[:, 0]], s1[1][px[:, 1]], s1[2][px[:, 2]]
    out = np.stack([s2[i][np.clip((M[i, 0] * R + M[i, 1] * G + M[i, 2] * B + 0x2000) >> 14, 0, 16384)] for i in range(3)], axis=1)
    assert np.array_equal(out, V[name + ".out8"].reshape(-1, 3))


@pytest.mark.parametrize("name", NAMES)
def test_16bit_table_matches_vectors(name):
    icc = _icc(name)
    t = pkg.IccClut16()
    assert pkg.load().avifgpu_icc_prepare_clut16(icc, len(icc), ctypes.byref(t)) == 0
    table = np.ctypeslib.as_array(t.table).reshape(33, 33, 33, 4)[..., :3]
    i = V[name + ".in16"].reshape(-1, 3).astype(np.int64)
    h2l = np.clip((((i.astype(np.float32) / np.float32(32768.0)) * np.float32(65535.0)) + np.float32(0.5)).astype(np.int64), 0, 65535)
    o = _tetrahedral(table, h2l)
    l2h = np.clip((((o.astype(np.float32) / np.float32(65535.0)) * np.float32(32768.0)) + np.float32(0.5)).astype(np.int64), 0, 32768)
    assert np.array_equal(l2h, V[name + ".out16"].reshape(-1, 3).astype(np.int64))


def _gpu_rows(gpu, d, src, icc):
    import torch
    dev = f"cuda:{gpu.device}"
    bufs = harness._alloc_write_out(d, d.height)
    d_src = torch.from_numpy(src.view(np.uint8).reshape(-1).copy()).to(dev)
    d_out = {pl: torch.from_numpy(b.view(np.uint8).reshape(-1).copy()).to(dev) for pl, b in bufs.items()}
    ptrs = [d_out[i].data_ptr() if i in d_out else None for i in range(4)]
    strides = [bufs[i].strides[0] if i in bufs else 0 for i in range(4)]
    gpu.write_rows(d, 0, d.height, d_src.data_ptr(), src.strides[0], ptrs, strides, mem=pkg.MEM_DEVICE,
                   stream=torch.cuda.current_stream(dev).cuda_stream, icc=icc)
    torch.cuda.synchronize(dev)
    for pl in bufs:
        bufs[pl] = d_out[pl].cpu().numpy().view(bufs[pl].dtype).reshape(bufs[pl].shape)
    return harness._trim(d, bufs, d.height, harness.write_planes)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_matches_vectors(gpu, name):
    icc = _icc(name)
    n = V[name + ".in8"].size // 3
    # 8-bit: fused ICC + 8-bit copy == lcms2's bytes
    d = pkg.WriteDesc(width=n, height=1, depth=8, planes=3, bit_depth=8, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
    got = _gpu_rows(gpu, d, V[name + ".in8"].copy(), gpu.icc_prepare_shaper8(icc))
    assert np.array_equal(got[0], V[name + ".out8"])
    # 16-bit: fused ICC + 12-bit rescale == lcms2's [0, 32768] result through the reference's rescale LUT
    d = pkg.WriteDesc(width=n, height=1, depth=16, planes=3, bit_depth=12, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
    got = _gpu_rows(gpu, d, V[name + ".in16"].copy(), gpu.icc_prepare_clut16(icc))
    assert np.array_equal(got[0], harness.oracle_write(d, V[name + ".out16"].copy())[0])
    # 32-bit (parametric curves only): HDR -> Rec.2020 + PQ, SDR -> sRGB + Clip; tier-2 bar
    if name + ".in32" in V.files:
        for key, target, transfer in ((".rec2020", pkg.ICC_TARGET_REC2020_LINEAR, pkg.TRANSFER_PQ), (".srgbf", pkg.ICC_TARGET_SRGB_FLOAT, pkg.TRANSFER_CLIP)):
            d = pkg.WriteDesc(width=n, height=1, depth=32, planes=3, bit_depth=12, transfer=transfer, peak_nits=1000,
                              alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
            got = _gpu_rows(gpu, d, V[name + ".in32"].copy(), gpu.icc_prepare(icc, target))
            want = harness.oracle_write(d, V[name + key].copy())
            st = harness.compare_write(d, want, got)
            assert st["max_abs"] <= 1 and st["exact_frac"] >= 0.985, (name, key, st)
