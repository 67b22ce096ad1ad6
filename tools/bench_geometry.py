#!/usr/bin/env python3
"""Streaming kernel vs generic kernel over a sweep of frame geometries (device-resident, 16-byte padded rows like libheif's).
   [CHROMA=444|422|420] python tools/bench_geometry.py [depth] [WxH ...]   depth 16 (RGB16 -> 12-bit) or 32 (RGB f32 -> 10-bit PQ)
"streaming" forces the streaming kernels at every size (tuning-word bit 3); the library's default takes the size-gated ones from 40 Mpx up."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 16
CH = {"444": (pkg.CHROMA_444, 0, 0), "422": (pkg.CHROMA_422, 1, 0), "420": (pkg.CHROMA_420, 1, 1)}[os.environ.get("CHROMA", "444")]
sizes = [tuple(map(int, a.split("x"))) for a in sys.argv[2:]] or [(8192, 8192), (6144, 4000), (6000, 4000), (6656, 4000), (4096, 4096), (7952, 5304), (8192, 5304)]
gpu = pkg.AvifGpu(0); dev = torch.device("cuda", 0)
def align(v, a): return (v + a - 1) // a * a
for (W, H) in sizes:
    bpp = 3 * depth // 8
    stride = align(W * bpp, 16)
    src = torch.randint(0, 120, (H, stride), dtype=torch.uint8, device=dev)
    d = pkg.WriteDesc(width=W, height=H, depth=depth, planes=3, bit_depth=12 if depth == 16 else 10,
                      transfer=pkg.TRANSFER_PQ if depth == 32 else pkg.TRANSFER_CLIP, peak_nits=80, alpha_state=pkg.ALPHA_NONE,
                      output=pkg.OUT_YCBCR, chroma=CH[0], matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
    pst = align(W * 2, 16); cst = align(((W + CH[1]) >> CH[1]) * 2, 16); ch = (H + CH[2]) >> CH[2]
    planes = [torch.empty((H, pst), dtype=torch.uint8, device=dev)] + [torch.empty((ch, cst), dtype=torch.uint8, device=dev) for _ in range(2)]
    ptrs = [t.data_ptr() for t in planes] + [None]; strides = [pst, cst, cst, 0]
    st = torch.cuda.current_stream(dev).cuda_stream
    row = {"geometry": f"{W}x{H}", "depth": depth}
    for name, variant in (("generic", 0), ("streaming", 7)):
        gpu.lib.avifgpu_set_hot_variant(variant | (8 if variant else 0))
        def step(): gpu.write_rows(d, 0, H, src.data_ptr(), stride, ptrs, strides, mem=pkg.MEM_DEVICE, stream=st)
        for _ in range(200): step()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); [step() for _ in range(100)]; b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 100)
        by = W * H * (bpp + 2 + 4.0 / ((1 << CH[1]) * (1 << CH[2])))
        row[name] = {"ms": round(best, 4), "frac": round(by / best / 1e6 / 8000, 3), "kernel": gpu.last_kernel()[:32]}
    gpu.lib.avifgpu_set_hot_variant(7)
    print(json.dumps(row), flush=True)
