#!/usr/bin/env python3
"""What the FIRST conversions of a process cost: wall time of the first launch of four kernel families (each lives in a code object of its own
since round 5: the HIP runtime loads an object on the first launch of one of its kernels), on a small tile so that the kernel itself is
negligible.  Run it several times: every process pays this once.      AVIFGPU_LIB=<library> python tools/first_launch.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev); torch.cuda.synchronize()        # the HIP context and torch's own first kernel are not ours to count
t0 = time.perf_counter(); gpu = pkg.AvifGpu(0); t_init = (time.perf_counter() - t0) * 1e3
W, H = 1024, 64
st = torch.cuda.Stream(dev)
out = []


def once(name, d, src, planes):
    ptrs = [p.data_ptr() for p in planes] + [None] * (4 - len(planes)); strides = [p.stride(0) for p in planes] + [0] * (4 - len(planes))
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        gpu.write_rows(d, 0, H, src.data_ptr(), src.stride(0) * src.element_size(), ptrs, strides, mem=pkg.MEM_DEVICE, stream=st.cuda_stream)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    out.append("%s first %.2f ms, then %.3f / %.3f" % (name, ts[0], ts[1], ts[2]))


f32 = torch.rand((H, W * 3), dtype=torch.float32, device=dev)
u16p = lambda n, w=W: [torch.empty((H, w * 2), dtype=torch.uint8, device=dev) for _ in range(n)]
once("RGB f32 -> PQ 4:4:4 (part 1: the headline kernel)", pkg.WriteDesc(width=W, height=H, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80, alpha_state=0, output=pkg.OUT_YCBCR,
                                             chroma=pkg.CHROMA_444, matrix_coefficients=9, color_primaries=9), f32, u16p(3))
u8 = torch.randint(0, 256, (H, W * 3), dtype=torch.uint8, device=dev)
once("RGB8 -> 8-bit 4:2:0 (part 2: 8/16-bit streaming)", pkg.WriteDesc(width=W, height=H, depth=8, planes=3, bit_depth=8, alpha_state=0, output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_420, matrix_coefficients=1),
     u8, [torch.empty((H, W), dtype=torch.uint8, device=dev), torch.empty((H // 2, W // 2), dtype=torch.uint8, device=dev), torch.empty((H // 2, W // 2), dtype=torch.uint8, device=dev)])
g32 = torch.rand((H, W), dtype=torch.float32, device=dev)
once("gray f32 -> PQ (part 32: generic, gray f32)", pkg.WriteDesc(width=W, height=H, depth=32, planes=1, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80, alpha_state=0, output=pkg.OUT_REFERENCE), g32, u16p(1))
print("avifgpu_init %.2f ms | " % t_init + " | ".join(out), flush=True)
