"""Why does bench.py's c3 row read 4-5 % below tools/bench_configs.py's C3 row?  The same row measured (a) by extra_configs in a FRESH process,
(b) after an allocation history like bench.py's (big buffers allocated and freed first), (c) with K = 20 / 60 / 200 timed launches."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
import importlib
pkg = importlib.import_module("avif-format_amd")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
gpu = pkg.AvifGpu(0)
stream = torch.cuda.current_stream(dev)
def show(tag, steps):
    r = bench.extra_configs(torch, pkg, gpu, dev, stream, steps)
    print(tag, {k: (r[k].get("ms"), r[k].get("frac")) for k in ("c2", "c3", "d8")}, flush=True)
show("fresh process, K=20", 20)
show("again, K=60", 60)
show("again, K=100", 100)
big = [torch.empty(int(2.1e9), dtype=torch.uint8, device=dev) for _ in range(4)]
del big
show("after 8.4 GB allocated and freed, K=20", 20)
torch.cuda.empty_cache()
show("after empty_cache, K=20", 20)
