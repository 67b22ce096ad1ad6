#!/usr/bin/env python3
"""How long does the C4 kernel need to reach its steady time after the process starts?  Batches of 20 back-to-back launches, one HIP
event pair per batch, for ~2 s from the first launch (after 1 s of idle)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda", 0)
gpu = pkg.AvifGpu(0)
W = H = 8192
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 10
d = pkg.WriteDesc(width=W, height=H, depth=32, planes=3, bit_depth=bits, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=pkg.CHROMA_444,
                  matrix_coefficients=9, color_primaries=9)
g = torch.Generator(device=dev); g.manual_seed(1)
src = torch.rand(H, W * 3, generator=g, device=dev, dtype=torch.float32)
planes = [torch.empty((H, W * 2), dtype=torch.uint8, device=dev) for _ in range(3)]
ptrs = [p.data_ptr() for p in planes] + [None]; strides = [p.stride(0) for p in planes] + [0]
st = torch.cuda.Stream(dev)
torch.cuda.synchronize(dev); time.sleep(1.0)
evs = []
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.5:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(20):
        gpu.write_rows(d, 0, H, src.data_ptr(), src.stride(0) * 4, ptrs, strides, mem=pkg.MEM_DEVICE, stream=st.cuda_stream)
    b.record(st)
    evs.append((time.perf_counter() - t0, a, b))
    if len(evs) % 8 == 0:
        torch.cuda.synchronize(dev)
torch.cuda.synchronize(dev)
ts = [(t, a.elapsed_time(b) / 20) for t, a, b in evs]
acc = 0.0
print(gpu.last_kernel())
for i in range(0, len(ts), max(1, len(ts) // 40)):
    chunk = ts[i:i + max(1, len(ts) // 40)]
    print("t=%.3f s  launches %5d..  mean %.4f ms" % (chunk[0][0], i * 20, sum(c[1] for c in chunk) / len(chunk)))
