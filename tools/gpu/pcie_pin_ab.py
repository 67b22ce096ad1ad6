#!/usr/bin/env python3
"""A/B of AVIFGPU_PIN_WORKERS (worker threads + pinned staging on the device's NUMA node) on the host-pointer path: C4 frame from
page-locked and from pageable caller memory, alternating the setting, best of 4 each, two rounds."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import harness  # noqa: E402

pkg = harness.pkg
W = H = 8192
d = pkg.WriteDesc(width=W, height=H, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80,
                  alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444,
                  matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
src = torch.rand((H, W * 3), dtype=torch.float32)
outs = [torch.empty((H, W * 2), dtype=torch.uint8) for _ in range(3)]
psrc = src.pin_memory()
pouts = [o.pin_memory() for o in outs]
for rnd in range(2):
    for pin in ("1", "0"):
        os.environ["AVIFGPU_PIN_WORKERS"] = pin
        pkg.load().avifgpu_shutdown()
        gpu = pkg.AvifGpu(0)
        for name, s, o in (("pinned", psrc, pouts), ("pageable", src, outs)):
            ptrs = [t.data_ptr() for t in o] + [None]
            strides = [t.stride(0) for t in o] + [0]
            best = None
            for _ in range(5):
                t0 = time.perf_counter()
                gpu.write_rows(d, 0, H, s.data_ptr(), s.stride(0) * 4, ptrs, strides, mem=pkg.MEM_HOST)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            print(json.dumps({"AVIFGPU_PIN_WORKERS": pin, "memory": name, "seconds": round(best, 5), "H2D_GB_s": round(W * H * 12 / best / 1e9, 1),
                              "topology": gpu.topology()}), flush=True)
