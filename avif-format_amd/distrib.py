"""One-process-per-GPU glue for bench.py (and its CPU test): ranks never exchange pixels -- row tiles / frames are
independent -- so torch.distributed only supplies the barrier around the timed region and the
MAX-over-ranks of its duration.  Backend "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests."""
from __future__ import annotations

import os
import time


class Ranks:
    def __init__(self, backend: str | None = None, device=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                try:
                    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
                    dist.init_process_group(backend=backend or "gloo", **kw)
                except Exception as exc:      # RCCL unavailable on this node: the barrier + MAX work over gloo just as well
                    if backend != "nccl":
                        raise
                    import sys
                    print(f"[distrib] nccl init failed ({exc}); using gloo for barrier/MAX", file=sys.stderr)
                    dist.init_process_group(backend="gloo")
            self.dist = dist
            # a CPU-side (gloo) group next to RCCL: ranks that only WAIT (rank 0's single-process sections of bench.py) then block
            # on a socket instead of spinning in a collective kernel on their GPU
            self.cpu_group = None
            if dist.get_backend() == "nccl":
                try:
                    self.cpu_group = dist.new_group(backend="gloo")
                except Exception:                # noqa: BLE001 -- fall back to the device barrier
                    self.cpu_group = None

    def host_barrier(self):
        """Barrier that keeps the GPUs idle while ranks wait (falls back to barrier())."""
        if self.dist is None:
            return
        if getattr(self, "cpu_group", None) is not None:
            self.dist.barrier(group=self.cpu_group)
        else:
            self.barrier()

    def barrier(self):
        if self.dist is not None:
            if self.dist.get_backend() == "nccl" and self.device is not None:
                self.dist.barrier(device_ids=[self.device.index if hasattr(self.device, "index") else int(self.device)])
            else:
                self.dist.barrier()

    def max_over_ranks(self, seconds: float) -> float:
        if self.dist is None:
            return seconds
        import torch
        dev = self.device if (self.device is not None and self.dist.get_backend() == "nccl") else "cpu"
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, step, steps: int, sync=lambda: None) -> float:
        """Barrier + sync, EXACTLY `steps` calls of step(), sync + barrier; returns MAX-over-ranks seconds."""
        self.barrier()
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        sync()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def gather_objects(self, obj):
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()
