import sys, json, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench, __graft_entry__ as e
pkg = e.load_package(); gpu = pkg.AvifGpu(0)
dev = torch.device("cuda", 0); stream = torch.cuda.Stream(dev)
for rep in range(2):
    out = bench.extra_configs(torch, pkg, gpu, dev, stream, 200)
    print({k: (v.get("ms"), v.get("frac")) for k, v in out.items()})
