#!/usr/bin/env python3
"""Is the C4 kernel's time a function of WHERE its buffers lie?  One pool; the source at offset 0, the three planes behind it at
controlled distances.  (The first row of a bench_configs process measured 2.7 % slower than the same row later in the same process.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda", 0)
gpu = pkg.AvifGpu(0)
W = H = 8192
d = pkg.WriteDesc(width=W, height=H, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=pkg.CHROMA_444,
                  matrix_coefficients=9, color_primaries=9)
SRC = W * H * 12; PL = W * H * 2
pool = torch.empty(SRC + 3 * PL + (256 << 20), dtype=torch.uint8, device=dev)
base = pool.data_ptr()
base += (-base) % (2 << 20)                                # 2 MiB aligned
g = torch.Generator(device=dev); g.manual_seed(1)
tmp = torch.rand(H * W * 3, generator=g, device=dev, dtype=torch.float32)
off0 = base - pool.data_ptr()
pool[off0:off0 + SRC].view(torch.float32).copy_(tmp); del tmp
st = torch.cuda.Stream(dev)
def run(poff, gap):
    ptrs = [base + SRC + poff + k * (PL + gap) for k in range(3)] + [None]
    strides = [W * 2] * 3 + [0]
    for _ in range(300):
        gpu.write_rows(d, 0, H, base, W * 12, ptrs, strides, mem=pkg.MEM_DEVICE, stream=st.cuda_stream)
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(200):
        gpu.write_rows(d, 0, H, base, W * 12, ptrs, strides, mem=pkg.MEM_DEVICE, stream=st.cuda_stream)
    b.record(st); torch.cuda.synchronize(dev)
    return a.elapsed_time(b) / 200
print(gpu.last_kernel() if False else "kernel timing vs plane placement")
for poff, gap in [(0, 0), (4096, 0), (65536, 0), (1 << 20, 0), (0, 4096), (0, 65536), (0, 1 << 20), (8192 + 256, 4096 + 256), (3 << 20, 5 << 20), (0, 0)]:
    print("plane0 at src_end + %8d, planes %8d apart: %.4f ms" % (poff, gap, run(poff, gap)))
