"""CPU: frame sharding + the scalar all-gather, world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from prisma_amd import shard


@pytest.mark.parametrize("n,world", [(256, 8), (10, 4), (3, 8), (1024, 8), (7, 2)])
def test_shards_partition_the_video(n, world):
    seen = []
    for r in range(world):
        s, e = shard.shard_range(n, r, world)
        seen += list(range(s, e))
    assert seen == list(range(n))
    s, e = shard.shard_range(n, world - 1, world, halo=1)
    s0, _ = shard.shard_range(n, world - 1, world)
    assert s == (s0 - 1 if e > s0 and s0 > 0 else s0)


def _worker(rank, world, port, n, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, e = shard.shard_range(n, rank, world)
    local = np.stack([np.arange(s, e, dtype=np.float32), 100.0 + np.arange(s, e, dtype=np.float32)], 1)
    out = shard.gather_frame_scalars(local, n)
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 8, 1])
def test_gather_frame_scalars_gloo_world2(n):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.shape == (n, 2)
    assert np.array_equal(out[:, 0], np.arange(n)) and np.array_equal(out[:, 1], 100 + np.arange(n))
