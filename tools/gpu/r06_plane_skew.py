"""Does the relative placement of the source and the three planes in memory matter to the headline kernel?  One arena per buffer set; the
source and the planes carved out of it at chosen byte skews from 2-MiB boundaries (torch hands large tensors out 2-MiB aligned: skew 0 is
what bench.py measures).  C4 (8192^2 RGB f32 -> 10-bit PQ 4:4:4), 4 rotating sets, K = 100 after a 0.15 s ramp."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib
import torch
import bench
pkg = importlib.import_module("avif-format_amd")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
gpu = pkg.AvifGpu(0)
stream = torch.cuda.current_stream(dev)
W = H = 8192
d = pkg.WriteDesc(width=W, height=H, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80, alpha_state=pkg.ALPHA_NONE,
                  output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020)
src0 = bench.make_frame(torch, dev, W, H, 3, 1234)
SRC, PL, MB2 = W * H * 12, W * H * 2, 2 << 20
def up(x): return (x + MB2 - 1) // MB2 * MB2
def run(skews, nset=4, K=100):
    calls, keep = [], []
    for _ in range(nset):
        arena = torch.empty(up(SRC) + 3 * up(PL) + 8 * MB2, dtype=torch.uint8, device=dev)
        base = (arena.data_ptr() + MB2 - 1) // MB2 * MB2
        s_ptr = base + skews[0]
        ptrs = [base + up(SRC) + MB2 + k * (up(PL) + MB2) + skews[1 + k] for k in range(3)]
        off = s_ptr - arena.data_ptr()
        arena[off:off + SRC].view(torch.float32).copy_(src0.view(-1))
        keep.append(arena)
        calls.append(lambda s_ptr=s_ptr, ptrs=ptrs: gpu.write_rows(d, 0, H, s_ptr, W * 12, ptrs + [None], [W * 2] * 3 + [0], mem=pkg.MEM_DEVICE, stream=stream.cuda_stream))
    torch.cuda.synchronize(dev)
    i = 0
    for _ in range(800):
        calls[i % nset](); i += 1
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(K):
        calls[i % nset](); i += 1
    e1.record(stream)
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / K
    return ms, 18.0 * W * H / ms / 1e6 / 8000
CASES = {"all at 2-MiB boundaries (what torch hands out)": (0, 0, 0, 0), "planes 4 KiB apart in phase": (0, 4096, 8192, 12288),
         "planes 256 B / 512 B / 768 B": (0, 256, 512, 768), "planes 64 KiB + 256 B steps": (0, 65792, 131584, 197376),
         "planes 683 KiB steps (thirds of 2 MiB)": (0, 699392, 1398784, 0), "source + 1 MiB, planes at boundaries": (1 << 20, 0, 0, 0),
         "everything skewed": (1 << 20, 349696, 1048832 + 349696, 1747968)}
for rep in range(2):
    for name, sk in CASES.items():
        ms, frac = run(sk)
        print(f"pass {rep + 1}  {name:48s} {ms:.4f} ms  {frac:.3f}  {gpu.last_kernel()[:40]}", flush=True)
