"""N > 1 path on CPU: two gloo ranks each own one even-row tile of a frame (exactly what bench.py --scaling strong and
the 8-GPU sharding do), no rank talks to another about pixels, and the tiles put back together are byte-identical
to the single-rank frame.  Also exercises bench.py's timing glue (barrier + MAX over ranks)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, chroma, q, use_gpu=False):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import harness
    pkg = harness.pkg
    ranks = pkg.distrib.Ranks(backend="gloo")
    assert ranks.world == world and ranks.rank == rank
    d = pkg.WriteDesc(width=70, height=37, depth=16, planes=4, bit_depth=10, alpha_state=pkg.ALPHA_PREMULTIPLIED,
                      output=pkg.OUT_YCBCR, chroma=chroma, matrix_coefficients=pkg.MATRIX_BT709)
    src = harness.make_write_source(d)                     # same seed on every rank = the same frame
    r0, n = pkg.sharding.row_tile(d.height, world, rank, even=True)
    calls = []
    gpu = None
    if use_gpu:
        import torch
        gpu = pkg.AvifGpu(rank % max(torch.cuda.device_count(), 1))    # one rank per GPU (ranks share it on a 1-GPU box)

    def step(i):
        calls.append(i)
        if gpu is not None:                                            # the PRODUCT converts this rank's tile
            step.out = harness.gpu_write(gpu, d, src, row0=r0, nrows=n, mem="host")
        else:                                                          # no GPU here: the checker stands in, the glue is what runs
            step.out = harness.oracle_write(d, src, row0=r0, nrows=n)
    elapsed = ranks.timed(step, steps=3)
    assert calls == [0, 1, 2] and elapsed > 0
    slow = ranks.max_over_ranks(10.0 if rank == 1 else 1.0)
    assert slow == 10.0
    tiles = ranks.gather_objects((r0, n, {k: v.tobytes() for k, v in step.out.items()}, {k: v.shape for k, v in step.out.items()}))
    if rank == 0:
        whole = harness.oracle_write(d, src)
        tiles.sort(key=lambda t: t[0])
        assert tiles[0][0] == 0 and sum(t[1] for t in tiles) == d.height
        for pl in whole:
            parts = [np.frombuffer(t[2][pl], dtype=whole[pl].dtype).reshape(t[3][pl]) for t in tiles if t[1]]
            assert np.array_equal(np.concatenate(parts, axis=0), whole[pl]), pl
        q.put("ok")
    ranks.close()


def _run_ranks(chroma, use_gpu):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, chroma, q, use_gpu)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"


@pytest.mark.parametrize("chroma", [1, 3])       # 4:2:0 and 4:4:4
def test_two_rank_tiles(chroma):
    _run_ranks(chroma, use_gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("chroma", [1, 3])
def test_two_rank_tiles_library(chroma):
    """Same two-rank run with libavifgpu.so converting each rank's tile (rank 0 still checks the assembled frame against the
    oracle's whole-frame result: the document is 16-bit, so bit-exact)."""
    _run_ranks(chroma, use_gpu=True)


def _nccl_fallback_worker(rank, world, port, q, errdir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.stderr = open(os.path.join(errdir, f"rank{rank}.err"), "w")
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    pkg = entry.load_package()
    ranks = pkg.distrib.Ranks(backend="nccl", device=None)            # no GPU here: RCCL cannot come up
    assert ranks.dist.get_backend() == "gloo" and ranks.fallback_reason
    assert ranks.max_over_ranks(float(rank + 1)) == float(world)
    got = ranks.gather_objects(rank)
    ranks.host_barrier()
    ranks.close()
    sys.stderr.flush()
    q.put((rank, got))


def test_nccl_request_without_rccl_falls_back_to_gloo_on_every_rank(tmp_path):
    """bench.py asks for backend "nccl"; where RCCL cannot come up (here: no GPU at all) EVERY rank moves to gloo together -- the vote of
    distrib.Ranks._init_nccl_or_gloo -- rank 0 says so in exactly ONE stderr line, and barrier / MAX / gather work."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_gpu_bench_nccl_branch.py covers the GPU outcome")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_fallback_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, [0, 1]), (1, [0, 1])]
    lines0 = [l for l in open(tmp_path / "rank0.err").read().splitlines() if l.startswith("[distrib]")]
    lines1 = [l for l in open(tmp_path / "rank1.err").read().splitlines() if l.startswith("[distrib]")]
    assert len(lines0) == 1 and "using gloo" in lines0[0] and lines1 == [], (lines0, lines1)


def test_row_tiles_partition():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    sh = entry.load_package().sharding
    for h in (1, 2, 7, 37, 4096, 8192, 16384, 32767):
        for world in (1, 2, 3, 4, 8):
            tiles = sh.all_tiles(h, world, even=True)
            assert tiles[0][0] == 0 and sum(n for _, n in tiles) == h
            for (a, n), (b, _) in zip(tiles[:-1], tiles[1:]):
                assert a + n == b
            assert all(r0 % 2 == 0 for r0, n in tiles if n)
            assert all(n % 2 == 0 for r0, n in tiles[:-1] if n)          # only the last tile may be odd
    assert sh.all_tiles(8192, 8) == [(i * 1024, 1024) for i in range(8)]
