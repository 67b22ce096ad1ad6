#!/usr/bin/env python3
"""Soak of the host-pointer scheduler: hundreds of conversions, both directions, page-locked and pageable memory, 1-4 contexts on
the visible devices, small sub-tiles so every slot, ticket and copy helper is exercised many times; every result compared with the
first one of its kind.  A hang (a lost ticket, a helper that never signals) shows as the timeout of the command that runs this."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import harness  # noqa: E402

pkg = harness.pkg
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(7)
ndev = max(torch.cuda.device_count(), 1)
wd = pkg.WriteDesc(width=1000, height=601, depth=32, planes=4, bit_depth=12, transfer=pkg.TRANSFER_PQ, peak_nits=1000,
                   alpha_state=pkg.ALPHA_STRAIGHT, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, matrix_coefficients=pkg.MATRIX_BT2020_NCL,
                   color_primaries=pkg.PRIMARIES_BT2020)
rd = pkg.ReadDesc(width=1030, height=517, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_420, bit_depth=10, depth=16,
                  alpha_state=pkg.ALPHA_PREMULTIPLIED, matrix_coefficients=pkg.MATRIX_BT709)
src = harness.make_write_source(wd, seed=3)
planes = harness.make_read_source(rd, seed=4, stride_pad=8)
want_w = want_r = None
t0 = time.time()
for i in range(rounds):
    os.environ["AVIFGPU_CHUNK_MB"] = str(int(rng.integers(1, 4)))
    os.environ["AVIFGPU_UPLOAD_DEPTH"] = str(int(rng.integers(0, 3)))
    os.environ["AVIFGPU_LANES"] = str(int(rng.integers(1, 4)))
    os.environ["AVIFGPU_SLOTS"] = str(int(rng.integers(2, 5)))
    os.environ["AVIFGPU_COPY_THREADS"] = str(int(rng.integers(0, 5)))
    n = int(rng.integers(1, 5))
    gpu = pkg.AvifGpu(devices=[k % ndev for k in range(n)]) if i % 7 == 0 or i == 0 else gpu
    for _ in range(3):
        got = harness.gpu_write(gpu, wd, src, mem="host", stride_pad=8, return_raw=True)
        if want_w is None:
            want_w = got
        for pl in want_w:
            assert np.array_equal(got[pl], want_w[pl]), ("write", i, pl)
        r = harness.gpu_read(gpu, rd, planes, mem="host")
        if want_r is None:
            want_r = r
        assert np.array_equal(r, want_r), ("read", i)
print(f"soak ok: {rounds} rounds x 3 x (write + read) in {time.time() - t0:.1f} s")
