"""CPU: the distributed glue of bench.py (barrier, max over ranks, scalar all-gather) with two and eight ranks over gloo."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      PRISMA_BENCH_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    import bench
    R = bench.Ranks()
    assert (R.rank, R.world, R.backend) == (rank, world, "gloo")
    R.barrier()
    slowest = R.max_over_ranks(1.0 + rank)                     # every rank must see the slowest rank's time
    scal = torch.full((3, 4), float(rank + 1))
    gathered = torch.empty((world, 3, 4))
    R.all_gather(gathered, scal)
    q.put((rank, slowest, gathered.clone()))
    R.close()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_gloo(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, slowest, gathered in got:
        assert slowest == float(world)
        for r in range(world):
            assert torch.equal(gathered[r], torch.full((3, 4), float(r + 1)))
