#!/usr/bin/env python3
"""The sampled-curve ICC kernel (icc=6: one table lookup per sample) against the CONTENT of the frame: does locality of the lookups
help or hurt?  8192^2 RGB f32 -> 10-bit PQ 4:4:4, five inputs, HIP-event time of 100 launches each."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import harness
pkg = harness.pkg
P = pkg
gpu = pkg.AvifGpu(0)
dev = torch.device("cuda", 0)
L = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle_icc.so"))
L.oracle_icc_make_profile.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_void_p, ctypes.c_uint32]
big = ctypes.create_string_buffer(1 << 18)
n = L.oracle_icc_make_profile(1, 3, 1024.0, big, len(big))
icc = gpu.icc_prepare_sampled(big.raw[:n])
W = H = 8192
d = pkg.WriteDesc(width=W, height=H, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_444,
                  matrix_coefficients=9, color_primaries=9)
g = torch.Generator(device=dev); g.manual_seed(5)
y = torch.linspace(0, 1, H, device=dev).view(-1, 1, 1); x = torch.linspace(0, 1, W, device=dev).view(1, -1, 1)
ph = torch.tensor([0.0, 2.1, 4.2], device=dev).view(1, 1, 3)
smooth = (0.5 + 0.45 * torch.sin(6.0 * x + 3.0 * y + ph) * torch.cos(2.0 * y - x)).reshape(H, -1).contiguous()
inputs = {"uniform random [0,1)": torch.rand((H, W * 3), generator=g, device=dev),
          "smooth": smooth,
          "smooth + noise 1e-3": (smooth + 1e-3 * torch.randn(smooth.shape, generator=g, device=dev)).clamp_(0, 1),
          "smooth + noise 2e-2": (smooth + 2e-2 * torch.randn(smooth.shape, generator=g, device=dev)).clamp_(0, 1),
          "constant 0.5": torch.full((H, W * 3), 0.5, device=dev)}
planes = [torch.empty((H, W * 2), dtype=torch.uint8, device=dev) for _ in range(3)]
ptrs = [p.data_ptr() for p in planes] + [None]; strides = [p.stride(0) for p in planes] + [0]
st = torch.cuda.Stream(dev)
for name, src in inputs.items():
    fn = lambda: gpu.write_rows(d, 0, H, src.data_ptr(), src.stride(0) * 4, ptrs, strides, mem=pkg.MEM_DEVICE, stream=st.cuda_stream, icc=icc)
    for _ in range(300): fn()
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(100): fn()
    b.record(st); torch.cuda.synchronize(dev)
    print(json.dumps({"input": name, "ms": round(a.elapsed_time(b) / 100, 4), "kernel": gpu.last_kernel()[-30:]}), flush=True)
