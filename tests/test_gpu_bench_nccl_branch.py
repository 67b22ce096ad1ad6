"""bench.py's `backend="nccl"` branch before the driver's 8-GPU node runs it: two ranks launched exactly like the
driver launches them (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...`) on the ONE GPU of this
box (AVIFGPU_BENCH_SHARE_DEVICE=nccl: ranks wrap around the visible devices but still ask for RCCL).  The documented outcome is one of two:
  * RCCL comes up with both ranks on one device: `rank_backend.backend == "nccl"`, no [distrib] line on stderr;
  * RCCL refuses (two ranks on one device) and EVERY rank falls back to gloo together (distrib.Ranks._init_nccl_or_gloo):
    `rank_backend.backend == "gloo"` with the reason, exactly ONE "[distrib] nccl unavailable" line on stderr.
Either way the bench line is printed once, carries `per_rank` for both ranks with their row tiles, and the timed region's MAX over ranks
is positive.  The NUMBERS of such a run mean nothing (two processes share one GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_asking_for_rccl_on_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AVIFGPU_BENCH_SHARE_DEVICE="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-c5", "--no-pcie",
           "--no-pattern", "--no-cold", "--clock-ramp-ms", "20", "--width", "4096", "--height", "4096"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["value"] > 0 and d["ms_per_step"] > 0
    per_rank = sorted(d["per_rank"], key=lambda p: p["rank"])
    assert [p["rank"] for p in per_rank] == [0, 1] and [p["rows"] for p in per_rank] == [2048, 2048], per_rank
    assert all(p["kernel_ms_mean"] > 0 for p in per_rank)
    rb = d["rank_backend"]
    notes = [l for l in r.stderr.splitlines() if l.startswith("[distrib]")]
    print("rank_backend:", rb, "| stderr notes:", notes)
    if rb["backend"] == "nccl":
        assert rb["fallback_reason"] is None and notes == []
    else:
        assert rb["backend"] == "gloo" and rb["fallback_reason"]
        assert len(notes) == 1 and "using gloo for barrier/MAX" in notes[0], notes
