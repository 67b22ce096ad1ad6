#!/usr/bin/env python3
"""What the x16 link itself does with the C4 job's bytes (805 MB up, 403 MB down, page-locked memory), no kernel, no library:
plain hipMemcpyAsync patterns through torch, to tell the link's ceiling from the scheduler's losses.

  up_only / down_only      one direction alone, one copy
  both_big                 the frame up and the three planes down at once, one copy each (bench.py's plain_copies figure)
  chunked S MiB x L lanes  the library's traffic shape: the frame in S-MiB row tiles dealt over L streams, each stream doing
                           up(tile) -> down(3 plane pieces of the tile) in order, like a slot's stream does around its kernel
One JSON line per pattern: seconds (best of 5), GB/s per direction."""
import json
import sys
import time

import torch

W = H = 8192
dev = torch.device("cuda", 0)
h_src = torch.empty((H, W * 3), dtype=torch.float32).pin_memory()
h_src.uniform_()
h_out = [torch.empty((H, W * 2), dtype=torch.uint8).pin_memory() for _ in range(3)]
d_in = torch.empty((H, W * 3), dtype=torch.float32, device=dev)
d_out = [torch.empty((H, W * 2), dtype=torch.uint8, device=dev) for _ in range(3)]
UP, DOWN = W * H * 12, W * H * 6


def best_of(fn, n=5):
    best = None
    for _ in range(n):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return best


def report(name, dt, up, down):
    print(json.dumps({"pattern": name, "seconds": round(dt, 5), "H2D_GB_s": round(up / dt / 1e9, 1) if up else None,
                      "D2H_GB_s": round(down / dt / 1e9, 1) if down else None}), flush=True)


s_up, s_dn = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def up_only():
    with torch.cuda.stream(s_up):
        d_in.copy_(h_src, non_blocking=True)


def down_only():
    with torch.cuda.stream(s_dn):
        for o, d in zip(h_out, d_out):
            o.copy_(d, non_blocking=True)


def both_big():
    up_only()
    down_only()


def chunked(mib, lanes, slots=4, separate_down_stream=False):
    rows = max(2, (mib << 20) // (W * 12) // 2 * 2)
    streams = [torch.cuda.Stream(dev) for _ in range(lanes * slots)]
    dstreams = [torch.cuda.Stream(dev) for _ in range(lanes * slots)] if separate_down_stream else streams

    def run():
        i = 0
        for r0 in range(0, H, rows):
            r1 = min(H, r0 + rows)
            st, ds = streams[i % len(streams)], dstreams[i % len(streams)]
            with torch.cuda.stream(st):
                d_in[r0:r1].copy_(h_src[r0:r1], non_blocking=True)
            if separate_down_stream:
                ds.wait_stream(st)
            with torch.cuda.stream(ds):
                for o, d in zip(h_out, d_out):
                    o[r0:r1].copy_(d[r0:r1], non_blocking=True)
            i += 1
    return run


report("up_only", best_of(up_only), UP, 0)
report("down_only", best_of(down_only), 0, DOWN)
report("both_big", best_of(both_big), UP, DOWN)
for mib in (8, 32, 128):
    for lanes in (1, 2):
        report(f"chunked {mib} MiB x {lanes} lanes x 4 slots", best_of(chunked(mib, lanes)), UP, DOWN)
report("chunked 8 MiB x 2 lanes, planes down on other streams", best_of(chunked(8, 2, separate_down_stream=True)), UP, DOWN)
report("chunked 8 MiB x 1 stream", best_of(chunked(8, 1, slots=1)), UP, DOWN)
report("chunked 8 MiB x 2 streams", best_of(chunked(8, 1, slots=2)), UP, DOWN)
