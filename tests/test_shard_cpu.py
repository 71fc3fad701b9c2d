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
    if rk.main:
        relay.drain_begin(n, chunk, write)       # bounded spool (PRISMA_SPOOL_MAX_CHUNKS): drains on a thread during the gather
    mx = rk.gather(np.arange(first, last, dtype=np.float32), n)
    if rk.main:
        relay.drain_end()
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


@pytest.mark.parametrize("world,n,chunk", [(2, 120, 4), (3, 50, 2)])
def test_bounded_spool_does_not_deadlock(tmp_path, monkeypatch, world, n, chunk):
    """with PRISMA_SPOOL_MAX_CHUNKS = 1 a rank > 0 blocks in put() until rank 0 consumes, and the scalar gather needs
    every rank to have finished - so rank 0 must drain WHILE it waits in the gather (Relay.drain_begin / drain_end).  15-25 chunks per
    shard against a bound of one: the old gather-then-drain order hangs here until the relay timeout."""
    monkeypatch.setenv("PRISMA_SPOOL_MAX_CHUNKS", "1")
    monkeypatch.setenv("PRISMA_RELAY_TIMEOUT_S", "60")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_relay_worker, args=(r, world, port, n, chunk, str(tmp_path / "band.npy"), q)) for r in range(world)]
    for p in procs:
        p.start()
    muxed, peak, _, spool, left, mx = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert muxed == [i % 251 for i in range(n)] and not left and mx == list(map(float, range(n)))


def _fallback_worker(rank, world, port, out_path, primary, q):
    import torch.distributed as dist      # noqa: F401
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      PRISMA_DIST_BACKEND="gloo", PRISMA_SPOOL=primary)
    rk = shard.Ranks()
    relay = shard.Relay(rk, out_path, timeout_s=60)
    got = []
    if not rk.main:
        relay.put(4, {"rgb": np.full((2, 8, 8, 3), 4, np.uint8)})              # lands in the primary spool
        import shutil
        shutil.rmtree(relay.dir)                                               # the tmpfs "fills up": the directory is gone -> OSError
        relay._mine = []
        relay.put(4, {"rgb": np.full((2, 8, 8, 3), 4, np.uint8)})              # -> falls back to the output's folder
        relay.put(6, {"rgb": np.full((2, 8, 8, 3), 6, np.uint8)})
        os.makedirs(relay.dir, exist_ok=True)                                  # for close()'s done file
    mx = rk.gather(np.zeros(4, np.float32), 8)
    if rk.main:
        relay.drain_begin(8, 2, lambda s, c: got.append((s, int(c["rgb"][0, 0, 0, 0]))))
        relay.drain_end()
    d1, d2 = relay.dir, relay.dir2
    relay.close()
    if rk.main:
        q.put((got, d1, d2, os.path.exists(d1), os.path.exists(d2)))
    rk.close()


def test_spool_falls_back_to_the_output_folder(tmp_path):
    """a chunk the primary spool refuses (ENOSPC on a small /dev/shm, or the directory vanished) goes to the run's fallback
    directory next to the output; rank 0 looks in both and removes both at the end."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    primary = tmp_path / "shm"
    primary.mkdir()
    (tmp_path / "out").mkdir()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, str(tmp_path / "out" / "band.npy"), str(primary), q)) for r in range(2)]
    for p in procs:
        p.start()
    got, d1, d2, left1, left2 = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == [(4, 4), (6, 6)]
    assert d1.startswith(str(primary)) and d2.startswith(str(tmp_path / "out")) and not left1 and not left2


def test_spool_base_and_stale_sweep(tmp_path, monkeypatch):
    """/dev/shm is only chosen when it has room for the estimate (Docker's default is 64 MB); spool directories of dead
    owners are swept, live or fresh ones are not."""
    import shutil
    out_dir = str(tmp_path)
    free = shutil.disk_usage("/dev/shm").free if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else 0
    if free:
        assert shard._pick_spool_base(out_dir, free * 2) == out_dir                    # too big for the tmpfs
        assert shard._pick_spool_base(out_dir, 1 << 20) == ("/dev/shm" if free > (80 << 20) else out_dir)
        monkeypatch.setenv("PRISMA_SPOOL_MIN_FREE", str(free * 2))
        assert shard._pick_spool_base(out_dir, 1 << 20) == out_dir
    # an owner file names host + pid; a directory without one (a run's fallback directory used to have none) or written on
    # another host is presumed alive until a day has passed; the age is the newest mtime INSIDE the directory (waiting chunks)
    import socket as _socket
    host = _socket.gethostname()
    names = ("dead", "live", "fresh", "noowner", "foreign", "busy", "ancient")
    d = {k: tmp_path / ("prisma_spool.a.%d" % i) for i, k in enumerate(names)}
    for v in d.values():
        v.mkdir()
    (d["dead"] / "owner.999999999").write_text("%s 999999999\n" % host)           # no such pid on this host
    (d["dead"] / "chunk_000000000.npz").write_text("x")
    (d["live"] / ("owner.%d" % os.getpid())).write_text("%s %d\n" % (host, os.getpid()))
    (d["foreign"] / "owner.999999999").write_text("some-other-host 999999999\n")
    (d["busy"] / "owner.999999999").write_text("%s 999999999\n" % host)
    (d["busy"] / "chunk_000000004.npz").write_text("x")                             # written just now: the directory is in use
    (d["ancient"] / "chunk_000000000.npz").write_text("x")                          # no owner, untouched for two days
    now = __import__("time").time()
    for k, age in (("dead", 3600), ("live", 3600), ("noowner", 3600), ("foreign", 3600), ("busy", 3600), ("ancient", 2 * 86400)):
        for f in [d[k]] + [e for e in d[k].iterdir() if not (k == "busy" and e.name.startswith("chunk"))]:
            os.utime(f, (now - age, now - age))
    shard._sweep_stale_spools(str(tmp_path))
    left = {k for k, v in d.items() if v.exists()}
    assert left == {"live", "fresh", "noowner", "foreign", "busy"}, left


def _abort_worker(rank, world, port, out_path, q):
    import torch.distributed as dist      # noqa: F401
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      PRISMA_DIST_BACKEND="gloo", PRISMA_SPOOL_MAX_CHUNKS="1")
    rk = shard.Ranks()
    relay = shard.Relay(rk, out_path, timeout_s=120)
    if rk.main:
        def bad_write(s, c):
            raise IOError("VideoWriter died at chunk %d" % s)
        relay.drain_begin(8, 2, bad_write)          # the drain thread fails on the first chunk it gets
        relay._thread.join(60)
        q.put(("main", type(relay._thread_err).__name__, os.path.exists(os.path.join(relay.dir, "abort"))))
    else:
        import time
        t0 = time.time()
        try:
            for s in (4, 6):                          # bound of one chunk: the second put() blocks until rank 0 consumes - or aborts
                relay.put(s, {"rgb": np.zeros((2, 4, 4, 3), np.uint8)})
            q.put(("producer", "no error", time.time() - t0))
        except RuntimeError as e:
            q.put(("producer", str(e), time.time() - t0))
    import shutil
    import torch.distributed as dist
    dist.barrier()
    if rk.main:
        shutil.rmtree(relay.dir, ignore_errors=True)
    dist.destroy_process_group()


def test_failed_drain_aborts_the_producers(tmp_path):
    """in bounded-spool mode a drain thread that raises used to leave the producers blocked in put() until the relay timeout
    (6 h by default); now it leaves an `abort` file that put() / close() poll."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_abort_worker, args=(r, 2, port, str(tmp_path / "band.npy"), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((k, (a, b)) for k, a, b in (q.get(timeout=120), q.get(timeout=120)))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got["main"] == ("OSError", True)
    assert "aborted the relay" in got["producer"][0] and "VideoWriter died" in got["producer"][0] and got["producer"][1] < 60


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


def test_owner_file_survives_concurrent_writers_and_a_dead_directory(tmp_path):
    """put()'s ENOSPC fallback calls _write_owner from every rank > 0 at once; a shared temporary name made the second os.replace
    raise inside put()'s except-handler.  Every writer now has its own temporary name and a failed write is not fatal."""
    import threading
    d = tmp_path / "fallback"
    d.mkdir()
    errs = []

    def work():
        try:
            for _ in range(50):
                shard._write_owner(str(d), 4242)
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=work) for _ in range(16)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert sorted(os.listdir(d)) == ["owner.4242"] and (d / "owner.4242").read_text().split()[1] == "4242"
    shard._write_owner(str(tmp_path / "does" / "not" / "exist"), 1)          # no directory: swallowed (the sweep treats "no owner" as unknown)


def test_inline_drain_failure_publishes_abort(tmp_path):
    """in the default unbounded mode drain_end() runs the drain inline; when it raises, the producers polling `done` in close()
    must find an `abort` file (in the spool, or in the fallback directory when the spool refuses it) instead of waiting out the timeout."""
    class Rk:
        world, rank, main = 2, 0, True
    for spool_ok in (True, False):
        relay = shard.Relay.__new__(shard.Relay)
        relay.rk = Rk()
        relay.timeout, relay.max_chunks = 5.0, 0
        relay.dir = str(tmp_path / ("spool%d" % spool_ok)) if spool_ok else str(tmp_path / "file_in_the_way" / "spool")
        relay.dir2 = str(tmp_path / ("fallback%d" % spool_ok))
        if spool_ok:
            os.makedirs(relay.dir)
        else:
            (tmp_path / "file_in_the_way").write_text("x")            # makedirs / open below it fail: the spool refuses the abort file
        relay._mine, relay._thread, relay._thread_err = [], None, None

        def bad_drain(*a):
            raise IOError("mux failed")
        relay.drain = bad_drain
        relay.drain_begin(4, 2, lambda s, c: None)
        with pytest.raises(IOError, match="mux failed"):
            relay.drain_end()
        where = relay.dir if spool_ok else relay.dir2
        assert "mux failed" in open(os.path.join(where, "abort")).read()
        assert os.path.isdir(relay.dir2) == (not spool_ok)
        producer = shard.Relay.__new__(shard.Relay)
        producer.dir, producer.dir2 = relay.dir, relay.dir2
        with pytest.raises(RuntimeError, match="aborted the relay"):
            producer._check_abort()
