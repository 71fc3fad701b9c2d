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


def _relay_worker(rank, world, port, n, chunk, out_path, q):
    """What a band's process_video does around the engine: chunks of encoded frames, the scalar gather, then rank 0 muxes the other
    ranks' chunks in frame order while they wait on the relay's file signal (no collective pending during the mux)."""
    import resource
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      PRISMA_DIST_BACKEND="gloo")
    rk = shard.Ranks()
    relay = shard.Relay(rk, out_path, timeout_s=120)
    first, last = rk.frames(n)
    H, W = 270, 480                                         # 389 KB per frame
    muxed, peak_held = [], 0
    frame = lambda i: np.full((H, W, 3), i % 251, np.uint8)
    for s in range(first, last, chunk):
        rgb = np.stack([frame(i) for i in range(s, min(last, s + chunk))])
        if rk.main:
            muxed += [int(f[0, 0, 0]) for f in rgb]
        else:
            relay.put(s, {"rgb": rgb})
        peak_held = max(peak_held, rgb.nbytes)

    def write(s, c):
        nonlocal peak_held
        peak_held = max(peak_held, c["rgb"].nbytes)
        muxed.extend(int(f[0, 0, 0]) for f in c["rgb"])
    mx = rk.gather(np.arange(first, last, dtype=np.float32), n)
    if rk.main:
        relay.drain(n, chunk, write)
    spool = relay.dir
    relay.close()
    if rk.main:
        q.put((muxed, peak_held, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss, spool, os.path.exists(spool), mx[:, 0].tolist()))
    rk.close()


@pytest.mark.parametrize("world,n,chunk", [(2, 200, 8), (8, 67, 4)])
def test_relay_delivers_chunks_in_frame_order_with_bounded_memory(tmp_path, world, n, chunk):
    """SURVEY 8(e) product path: 2 ranks x 200 frames of 389 KB (78 MB of video) and 8 ranks x 67 frames (ragged tail shard): rank 0
    receives every other rank's chunks in frame order through the spool, never holds more than one chunk of frames, the spool is a
    per-run directory (not next to the output when /dev/shm exists) that is gone at the end, and only the per-frame scalars went
    through the collective."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    out = str(tmp_path / "band.npy")
    procs = [ctx.Process(target=_relay_worker, args=(r, world, port, n, chunk, out, q)) for r in range(world)]
    for p in procs:
        p.start()
    muxed, peak, rss_kb, spool, spool_left, mx = q.get(timeout=300)
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert muxed == [i % 251 for i in range(n)]
    assert peak <= chunk * 270 * 480 * 3
    assert "prisma_spool.band.npy." in spool and not spool_left
    if os.path.isdir("/dev/shm") and not os.environ.get("PRISMA_SPOOL"):
        assert spool.startswith("/dev/shm/")
    assert mx == list(map(float, range(n)))


def test_two_concurrent_relays_do_not_share_a_spool(tmp_path, monkeypatch):
    """two jobs with the same output name and the same PRISMA_SPOOL used to share `chunk_*` names, and rank 0 of one deleted
    the other's files as leftovers.  Spool directories are per run now."""
    monkeypatch.setenv("PRISMA_SPOOL", str(tmp_path))
    ctx = mp.get_context("spawn")
    jobs = []
    for j in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        q = ctx.Queue()
        procs = [ctx.Process(target=_relay_worker, args=(r, 2, port, 40, 4, str(tmp_path / "band.npy"), q)) for r in range(2)]
        for p in procs:
            p.start()
        jobs.append((q, procs))
    spools = []
    for q, procs in jobs:
        muxed, _, _, spool, left, mx = q.get(timeout=300)
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        assert muxed == [i % 251 for i in range(40)] and not left and mx == list(map(float, range(40)))
        spools.append(spool)
    assert spools[0] != spools[1] and all(s.startswith(str(tmp_path)) for s in spools)


def test_cached_weights_equal_generated_ones(tmp_path, monkeypatch):
    """bench.py's multi-rank start-up: rank 0 generates the seeded weights once, the others memory-map them (synth.cached_weights)."""
    from prisma_amd import synth
    monkeypatch.setenv("PRISMA_SYNTH_CACHE", str(tmp_path))
    first = synth.cached_weights("raft", 4321)               # generated + stored
    again = synth.cached_weights("raft", 4321)               # memory-mapped
    ref = synth.raft_weights(4321)
    assert list(first) == list(again) == list(ref)
    assert all(np.array_equal(np.asarray(again[k]), ref[k]) and np.asarray(again[k]).dtype == ref[k].dtype for k in ref)
    assert isinstance(again["fnet.conv1.weight"], np.memmap)
    small = synth.cached_weights("depth", synth.DEPTH_CFGS["vits"], 7)
    assert np.array_equal(np.asarray(synth.cached_weights("depth", synth.DEPTH_CFGS["vits"], 7)["pretrained.cls_token"]), small["pretrained.cls_token"])
