#!/usr/bin/env python3
"""Row-tile size vs time per launch of the C4 kernel, and the host cost of issuing one avifgpu_write_rows(MEM_DEVICE) call through
ctypes: what a rank of the N-way row split sees (N = 8 -> 1024-row tiles).  python tools/launch_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, __graft_entry__ as entry
pkg = entry.load_package()
gpu = pkg.AvifGpu(0); dev = torch.device("cuda", 0)
W, H = 8192, 8192
frame = torch.rand((H, W * 3), dtype=torch.float32, device=dev)
d = pkg.WriteDesc(width=W, height=H, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80, alpha_state=pkg.ALPHA_NONE,
                  output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
planes = [torch.empty((H, W * 2), dtype=torch.uint8, device=dev) for _ in range(3)]
stream = torch.cuda.Stream(dev)
for nrows in (8192, 4096, 2048, 1024, 512, 64):
    ptrs = [p.data_ptr() for p in planes] + [None]; strides = [W * 2] * 3 + [0]
    def step(): gpu.write_rows(d, 0, nrows, frame.data_ptr(), W * 12, ptrs, strides, mem=pkg.MEM_DEVICE, stream=stream.cuda_stream)
    for _ in range(300): step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record(stream)
    for _ in range(400): step()
    b.record(stream); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"tile {nrows:5d} rows: GPU {a.elapsed_time(b)/400*1e3:7.1f} us/launch, host issue {(t1-t0)/400*1e6:6.1f} us/call, wall {(t2-t0)/400*1e6:7.1f} us/step", flush=True)
