import sys, os, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda:0")
gpu = pkg.AvifGpu(0)
W = H = 8192
def desc(ev):
    return pkg.WriteDesc(width=W, height=H, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80, alpha_state=pkg.ALPHA_NONE,
                         output=pkg.OUT_YCBCR, chroma=pkg.CHROMA_444, matrix_coefficients=pkg.MATRIX_BT2020_NCL, color_primaries=pkg.PRIMARIES_BT2020, pq_evaluation=ev)
sets = []
for i in range(4):
    src = torch.rand(H, W * 3, device=dev, dtype=torch.float32)
    planes = [torch.empty(H, W, device=dev, dtype=torch.int16) for _ in range(3)]
    sets.append((src, planes))
st = torch.cuda.current_stream()
def run(d, n):
    for i in range(n):
        s, p = sets[i % 4]
        gpu.write_rows(d, 0, H, s.data_ptr(), s.stride(0) * 4, [q.data_ptr() for q in p], [q.stride(0) * 2 for q in p], mem=pkg.MEM_DEVICE, stream=st.cuda_stream)
for rep in range(3):
    for name, ev in (("auto", pkg.PQ_AUTO), ("compact", pkg.PQ_COMPACT), ("close", pkg.PQ_CLOSE)):
        d = desc(ev)
        run(d, 300); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(d, 200); b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 200
        print(rep, name, "%.4f ms  %.3f of 8 TB/s" % (ms, W * H * 18 / ms / 1e6 / 8000), gpu.last_kernel(), flush=True)
