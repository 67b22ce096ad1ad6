#!/usr/bin/env python3
"""PCIe-inclusive timing of the host-pointer entry (avifgpu_write_rows, AVIFGPU_MEM_HOST): C4 (8192^2 RGB f32 -> 10-bit PQ
YCbCr 4:4:4: 805 MB in, 403 MB out) from page-locked and from pageable caller memory, for 1..N bound contexts, sub-tile sizes
and slot counts.  On a 1-GPU box the N contexts share one device and one x16 link: what scales there is the workers' bounce
memcpy of pageable buffers, not the DMA.  One JSON line per configuration."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import harness  # noqa: E402

pkg = harness.pkg


def run(W, H, ncontexts, chunk_mb, pinned, reps=4, slots=None):
    os.environ["AVIFGPU_CHUNK_MB"] = str(chunk_mb)
    if slots:
        os.environ["AVIFGPU_SLOTS"] = str(slots)
    ndev = torch.cuda.device_count()
    if slots:
        pkg.load().avifgpu_shutdown()          # slot count is read when contexts are created
    gpu = pkg.AvifGpu(devices=[i % ndev for i in range(ncontexts)])
    d = pkg.WriteDesc(width=W, height=H, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                      alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                      matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
    src = torch.rand((H, W * 3), dtype=torch.float32)
    outs = [torch.empty((H, W * 2), dtype=torch.uint8) for _ in range(3)]
    if pinned:
        src = src.pin_memory()
        outs = [o.pin_memory() for o in outs]
    ptrs = [o.data_ptr() for o in outs] + [None]
    strides = [o.stride(0) for o in outs] + [0]
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        gpu.write_rows(d, 0, H, src.data_ptr(), src.stride(0) * 4, ptrs, strides, mem=pkg.MEM_HOST)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(json.dumps({"config": f"{W}x{H} RGB f32 -> 10-bit PQ YCbCr 4:4:4, host pointers", "memory": "pinned" if pinned else "pageable",
                      "contexts": ncontexts, "physical_gpus": min(ncontexts, ndev), "chunk_MiB": chunk_mb,
                      "slots": int(os.environ.get("AVIFGPU_SLOTS", "4")), "seconds": round(best, 4),
                      "Mpx_s": round(W * H / best / 1e6, 1), "H2D_GB_s": round(W * H * 12 / best / 1e9, 1),
                      "total_GB_s": round(W * H * 18 / best / 1e9, 1)}), flush=True)


def run_read(W, H, chunk_mb, pinned, reps=4):
    """The open direction of the same frame: 10-bit YCbCr 4:4:4 planes (403 MB) up, RGB f32 rows (805 MB) down."""
    os.environ["AVIFGPU_CHUNK_MB"] = str(chunk_mb)
    gpu = pkg.AvifGpu(devices=[0])
    d = pkg.ReadDesc(width=W, height=H, colorspace=pkg.COLORSPACE_YCBCR, chroma=pkg.CHROMA_444, bit_depth=10, depth=32,
                     alpha_state=pkg.ALPHA_NONE, color_primaries=pkg.PRIMARIES_BT2020, transfer_characteristics=pkg.TC_PQ,
                     matrix_coefficients=pkg.MATRIX_BT2020_NCL, full_range_flag=1, pq_peak_nits=80)
    planes = [(torch.rand((H, W)) * 1023).to(torch.int16) for _ in range(3)]
    dst = torch.empty((H, W * 3), dtype=torch.float32)
    if pinned:
        planes = [p.pin_memory() for p in planes]
        dst = dst.pin_memory()
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        gpu.read_rows(d, 0, H, [p.data_ptr() for p in planes] + [None], [p.stride(0) * 2 for p in planes] + [0], dst.data_ptr(),
                      dst.stride(0) * 4, mem=pkg.MEM_HOST)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(json.dumps({"config": f"{W}x{H} 10-bit PQ YCbCr 4:4:4 -> RGB f32, host pointers", "memory": "pinned" if pinned else "pageable",
                      "chunk_MiB": chunk_mb, "seconds": round(best, 4), "Mpx_s": round(W * H / best / 1e6, 1),
                      "D2H_GB_s": round(W * H * 12 / best / 1e9, 1), "H2D_GB_s": round(W * H * 6 / best / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    W = H = 8192
    for chunk in (8, 32, 128):
        run(W, H, 1, chunk, True)
    for slots in (2, 3, 6):
        run(W, H, 1, 32, True, slots=slots)
    os.environ["AVIFGPU_SLOTS"] = "4"
    pkg.load().avifgpu_shutdown()
    for n in (1, 2, 4):
        run(W, H, n, 32, False)
    for n in (2, 4):
        run(W, H, n, 32, True)
    pkg.AvifGpu(0)
