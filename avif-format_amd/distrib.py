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
        self.fallback_reason = None
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                if backend == "nccl":
                    self._init_nccl_or_gloo(dist, device)
                else:
                    dist.init_process_group(backend=backend or "gloo")
            self.dist = dist
            # a CPU-side (gloo) group next to RCCL: ranks that only WAIT (rank 0's single-process sections of bench.py) then block
            # on a socket instead of spinning in a collective kernel on their GPU
            self.cpu_group = None
            if dist.get_backend() == "nccl":
                try:
                    self.cpu_group = dist.new_group(backend="gloo")
                except Exception:                # noqa: BLE001 -- fall back to the device barrier
                    self.cpu_group = None

    def _init_nccl_or_gloo(self, dist, device):
        """RCCL for the barrier + MAX, or -- when RCCL cannot come up on this node (no fabric, two ranks on one GPU in a rehearsal,
        a broken install) -- gloo, decided by ALL ranks together: every rank initialises the nccl group on one rendezvous store, runs ONE
        probe collective (communicators are created lazily: an init that "succeeds" proves nothing), votes through the store, and if any
        rank failed every rank tears the group down and meets again over gloo.  Rank 0 prints exactly one line on stderr when that
        happens.  No pixel ever crosses this group either way."""
        import datetime
        import sys
        import torch
        store, rank, world = next(iter(dist.rendezvous("env://", self.rank, self.world)))
        store.set_timeout(datetime.timedelta(seconds=120))
        reason = ""
        try:
            kw = {"device_id": device} if device is not None else {}
            dist.init_process_group(backend="nccl", store=dist.PrefixStore("nccl", store), rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=90), **kw)
            idx = device.index if hasattr(device, "index") else (int(device) if device is not None else None)
            dist.barrier(device_ids=[idx] if idx is not None else None)
            t = torch.ones(1, device=device if device is not None else "cuda")
            dist.all_reduce(t)
            if int(t.item()) != world:
                raise RuntimeError(f"probe all-reduce returned {t.item()} for world {world}")
        except Exception as exc:      # noqa: BLE001
            reason = (str(exc).strip().splitlines() or [type(exc).__name__])[0][:200]
        store.set(f"avifgpu_nccl_vote_{rank}", reason or "ok")
        votes = [store.get(f"avifgpu_nccl_vote_{r}").decode() for r in range(world)]
        if all(v == "ok" for v in votes):
            return
        if dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:        # noqa: BLE001 -- a half-built communicator; the gloo group below does not depend on it
                pass
        self.fallback_reason = next(v for v in votes if v != "ok")
        if rank == 0:
            print(f"[distrib] nccl unavailable ({self.fallback_reason}); using gloo for barrier/MAX", file=sys.stderr, flush=True)
        dist.init_process_group(backend="gloo", store=dist.PrefixStore("gloo", store), rank=rank, world_size=world)

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
