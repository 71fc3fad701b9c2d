"""Multi-GPU plumbing of the bands: static contiguous frame shards, one scalar all-gather, and an ordered per-chunk
relay of the encoded frames to rank 0 (the ranks of ONE node share a filesystem; no whole-video tensor exists anywhere).

The reference processes one frame at a time in one process (bands/depth_anything.py:203-225) and
carries no cross-frame state for depth (min/max normalisation is per frame, :215-217), so frames
shard embarrassingly.  Each rank runs its own pb_ctx on its own GPU; the only exchange is the
per-frame (min, max) pair that rank 0 needs, in frame order, for <band>_min.csv / _max.csv
(:232-238).  torch.distributed is the transport ("nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests); payload = 8 bytes per frame, one collective per video, so the default algorithm is
fine - xGMI bandwidth is irrelevant at this size.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def shard_range(n_frames: int, rank: int, world: int, halo: int = 0) -> Tuple[int, int]:
    """[start, stop) of the contiguous block of ceil(n/world) frames owned by `rank`.

    halo > 0 extends the start backwards (flow needs frame i-1 for pair i: bands/flow_raft.py:103-113).
    """
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    per = -(-n_frames // world)
    start = min(rank * per, n_frames)
    stop = min(start + per, n_frames)
    return max(0, start - halo) if stop > start else start, stop


def gather_rows(local: np.ndarray, n_frames: int, device=None) -> np.ndarray | None:
    """All-gather per-frame rows ([n_local, ...], any dtype torch knows) into frame order; rank 0 gets [n_frames, ...].

    Used for the per-frame scalars only (float32 min / max / max-displacement, 4-12 bytes per frame); the encoded frames
    reach rank 0 through Relay.  Shards have ceil(n/world) frames except the tail; every rank pads to that length so one
    fixed-size all_gather_into_tensor suffices.
    """
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local)
    if local.ndim == 1:
        local = local[:, None]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local[:n_frames]
    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-n_frames // world)
    tail = local.shape[1:]
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    buf = torch.zeros((per,) + tail, dtype=torch.from_numpy(local[:0]).dtype, device=dev)
    buf[: local.shape[0]] = torch.from_numpy(local).to(dev)
    out = torch.empty((world * per,) + tail, dtype=buf.dtype, device=dev)
    dist.all_gather_into_tensor(out, buf)
    if rank != 0:
        return None
    rows: List[np.ndarray] = []
    o = out.cpu().numpy()
    for r in range(world):
        s, e = shard_range(n_frames, r, world)
        rows.append(o[r * per: r * per + (e - s)])
    return np.concatenate(rows, 0)


def gather_frame_scalars(local: np.ndarray, n_frames: int, device=None) -> np.ndarray | None:
    """float32 rows ([n_local, k]): the (min, max) / max-displacement columns of the CSV files."""
    return gather_rows(np.asarray(local, np.float32), n_frames, device)


def synthetic_allowed(flag: bool = False) -> bool:
    """Seeded synthetic weights are for tests and benchmarks only: a band falls back to them only when asked to
    (`--synthetic` or PRISMA_SYNTH=1); otherwise a missing checkpoint is an error, never silent garbage in a PRISMA folder."""
    import os
    return bool(flag) or os.environ.get("PRISMA_SYNTH", "") == "1"


class Relay:
    """Ordered delivery of encoded chunks from every rank to rank 0, which muxes the output video(s).

    north_star allows a collective only for the per-frame scalars, so the frames do not travel through torch.distributed at
    all: the ranks are the GPUs of one node, and a rank > 0 drops each finished chunk ({name: array}, typically 16-32 encoded
    frames) as one file into a spool directory (write to a temporary name, then rename: the consumer never sees a partial
    file).  Rank 0 writes its own chunks straight to the video as they finish and then walks the other ranks' chunk plans in
    frame order, waiting for each file, muxing it and deleting it.  Memory is bounded by one chunk per rank.

    The spool is PER RUN: `<base>/prisma_spool.<basename(out)>.<token>` where the token is drawn by rank 0 and broadcast, so two
    jobs (or two bands of one job) never share a namespace and nobody deletes anybody else's files.  It holds, at most, the other
    ranks' encoded shards until rank 0 gets to them - (world - 1) / world of the uncompressed encoded video when nothing bounds it.
    `base`: PRISMA_SPOOL if set; else /dev/shm when it is writable AND has room for `est_bytes` (what the caller
    expects the other ranks to spool: 8 ranks x ~200 frames/s x 6.2 MB of 1080p frames is ~10 GB/s - memory, not the output
    filesystem - but Docker's default /dev/shm is 64 MB and a bare-metal tmpfs competes with pinned buffers); else the output's
    folder.  That folder is also every run's FALLBACK directory: a chunk that cannot be written to the primary spool (ENOSPC: the
    estimate was wrong, or another job filled the tmpfs) goes there instead, and rank 0 looks in both.  Spool directories whose
    owner (rank 0's pid, recorded in the directory) is dead are swept when a new run picks the same base.

    With PRISMA_SPOOL_MAX_CHUNKS = K > 0 a producer waits while K of its chunks are unconsumed (bounded spool, but the producers
    then finish one after the other at rank 0's mux speed: the default 0 keeps every GPU computing).

    End of a run: `drain_begin` / scalar gather / `drain_end`.  Unbounded (K = 0): the gather comes first - every rank
    reaches it as soon as its own compute is done - and rank 0 drains afterwards while the ranks > 0 wait for its `done` FILE
    (close), not inside a collective.  Bounded (K > 0): the producers cannot finish - and so cannot reach the gather -
    unless rank 0 consumes, so rank 0 drains on a worker thread WHILE it waits in the gather; draining only after the gather, as the
    unbounded order does, would deadlock as soon as a shard has more than K (+ the sink's depth) chunks."""

    def __init__(self, ranks: "Ranks", out_path: str, timeout_s: float = 0.0, est_bytes: int = 0):
        import os
        self.rk = ranks
        self.timeout = timeout_s or float(os.environ.get("PRISMA_RELAY_TIMEOUT_S", "21600"))
        self.max_chunks = int(os.environ.get("PRISMA_SPOOL_MAX_CHUNKS", "0"))
        self.dir = self.dir2 = ""
        self._mine: List[str] = []
        self._thread = None
        self._thread_err = None
        self._pending = None
        if ranks.world > 1:
            import uuid
            import torch.distributed as dist
            box = [None]
            if ranks.main:
                out_dir = os.path.dirname(os.path.abspath(out_path))
                base = os.environ.get("PRISMA_SPOOL") or _pick_spool_base(out_dir, est_bytes)
                name = "prisma_spool.%s.%s" % (os.path.basename(out_path), uuid.uuid4().hex[:12])
                box = [(os.path.join(base, name), os.path.join(out_dir, name), os.getpid())]
                _sweep_stale_spools(base)
                if out_dir != base:
                    _sweep_stale_spools(out_dir)
            dist.broadcast_object_list(box, src=0)
            self.dir, self.dir2, owner = box[0]
            os.makedirs(self.dir, exist_ok=True)
            self._owner = owner
            if ranks.main:
                _write_owner(self.dir, owner)

    def _path(self, start: int, fallback: bool = False) -> str:
        import os
        return os.path.join(self.dir2 if fallback else self.dir, "chunk_%09d.npz" % start)

    def _write(self, path: str, arrays: dict):
        import os
        tmp = path + ".tmp.%d" % self.rk.rank
        try:
            with open(tmp, "wb") as f:
                np.savez(f, **{k: np.ascontiguousarray(v) for k, v in arrays.items()})
            os.replace(tmp, path)
        except BaseException:
            try:
                os.remove(tmp)
            except OSError:
                pass
            raise

    def _check_abort(self):
        """rank 0's drain failed: it left an `abort` file so that producers blocked in put() / close() fail with its message
        instead of waiting out the relay timeout while rank 0 sits in the gather."""
        import os
        for d in dict.fromkeys((self.dir, self.dir2)):
            p = os.path.join(d, "abort")
            if os.path.exists(p):
                try:
                    msg = open(p).read().strip()
                except OSError:
                    msg = "?"
                raise RuntimeError(f"rank 0 aborted the relay: {msg}")

    def put(self, start: int, arrays: dict):
        """rank > 0: publish the chunk whose first unit (frame / pair index) is `start`."""
        import os
        import time
        self._check_abort()
        if self.max_chunks > 0:
            t0 = time.time()
            while sum(os.path.exists(p) for p in self._mine) >= self.max_chunks:
                if time.time() - t0 > self.timeout:
                    raise TimeoutError(f"rank 0 did not consume rank {self.rk.rank}'s chunks within {self.timeout:.0f} s")
                self._check_abort()
                time.sleep(0.005)
            self._mine = [p for p in self._mine if os.path.exists(p)]
        path = self._path(start)
        try:
            self._write(path, arrays)
        except OSError as e:
            # the primary spool (a tmpfs, usually) is full or gone: this chunk goes to the output's folder, where rank 0 also looks
            if self.dir2 == self.dir:
                raise
            import sys
            print(f"[prisma] rank {self.rk.rank}: spool {self.dir} refused chunk {start} ({e}); falling back to {self.dir2}", file=sys.stderr)
            os.makedirs(self.dir2, exist_ok=True)
            _write_owner(self.dir2, self._owner)       # the fallback directory names its owner too: a sweep must not take it for a dead run's
            path = self._path(start, True)
            self._write(path, arrays)
        self._mine.append(path)

    def drain(self, n_units: int, chunk: int, write):
        """rank 0: for every other rank, in rank (= frame) order, for every chunk start of its shard: wait for the file (primary
        spool or the fallback directory), call write(start, {name: array}), delete it."""
        import os
        import time
        for r in range(1, self.rk.world):
            first, last = shard_range(n_units, r, self.rk.world)
            for s in range(first, last, chunk):
                cand, t0 = (self._path(s), self._path(s, True)), time.time()
                while True:
                    p = next((c for c in cand if os.path.exists(c)), None)
                    if p:
                        break
                    if time.time() - t0 > self.timeout:
                        raise TimeoutError(f"rank {r} did not deliver chunk {s} within {self.timeout:.0f} s")
                    time.sleep(0.005)
                with np.load(p) as z:
                    write(s, {k: z[k] for k in z.files})
                os.remove(p)

    def drain_begin(self, n_units: int, chunk: int, write):
        """rank 0, BEFORE the scalar gather.  Bounded spool: the drain starts now, on a worker thread (see the class docstring);
        unbounded: it is only recorded and runs in drain_end, after the gather."""
        self._pending = (n_units, chunk, write)
        if self.max_chunks > 0:
            import threading

            def work():
                try:
                    self.drain(*self._pending)
                except BaseException as e:      # noqa: BLE001 - re-raised by drain_end
                    self._thread_err = e
                    self._abort(e)              # producers blocked in put() must not wait for a consumer that is gone
            self._thread = threading.Thread(target=work, name="relay-drain", daemon=True)
            self._thread.start()

    def _abort(self, err):
        """leave an `abort` file for the producers - in the spool and, when that one refuses (a full tmpfs), in the fallback directory"""
        import os
        import sys
        wrote = False
        for d in dict.fromkeys((self.dir, self.dir2)):
            try:
                os.makedirs(d, exist_ok=True)
                tmp = os.path.join(d, "abort.tmp")
                with open(tmp, "w") as f:
                    f.write("%s: %s\n" % (type(err).__name__, err))
                os.replace(tmp, os.path.join(d, "abort"))
                wrote = True
                break                           # the fallback directory is only used (and created) when the spool itself refuses the file
            except OSError:
                continue
        if not wrote:
            print(f"[prisma] rank 0: could not publish the relay's abort file ({type(err).__name__}: {err})", file=sys.stderr)

    def drain_end(self):
        """rank 0, AFTER the scalar gather: finishes (bounded) or runs (unbounded) the drain."""
        if self._thread is not None:
            self._thread.join()
            self._thread = None
            if self._thread_err is not None:
                err, self._thread_err = self._thread_err, None
                raise err
        elif self._pending is not None:
            pending, self._pending = self._pending, None
            try:
                self.drain(*pending)
            except BaseException as e:      # noqa: BLE001 - the producers poll `done` for hours otherwise: tell them, then re-raise
                self._abort(e)
                raise
        self._pending = None

    def close(self):
        """rank 0: publish `done`; ranks > 0: wait for it (a file poll, no collective in flight), then the last one out removes the
        directory."""
        import os
        import time
        if self.rk.world <= 1:
            return
        done = os.path.join(self.dir, "done")
        if self.rk.main:
            with open(done + ".tmp", "w") as f:
                f.write("ok\n")
            os.replace(done + ".tmp", done)
        else:
            t0 = time.time()
            while not os.path.exists(done):
                if time.time() - t0 > self.timeout:
                    raise TimeoutError(f"rank 0 did not finish muxing within {self.timeout:.0f} s")
                self._check_abort()
                time.sleep(0.02)
        import torch.distributed as dist
        dist.barrier()                       # short: every rank is past its file wait
        if self.rk.main:
            import shutil
            shutil.rmtree(self.dir, ignore_errors=True)
            if self.dir2 != self.dir and os.path.isdir(self.dir2):
                shutil.rmtree(self.dir2, ignore_errors=True)


def _write_owner(d: str, pid: int):
    """owner.<pid> holding `<host> <pid>`: who may be asked whether this spool is still in use (_sweep_stale_spools)."""
    import os
    import socket
    import uuid
    p = os.path.join(d, "owner.%d" % pid)
    if os.path.exists(p):
        return
    # several ranks may get here at once (a full tmpfs sends every producer to the fallback directory in the same moment): each writes through
    # its OWN temporary name, and a failure is not fatal - a directory without an owner file is "unknown, keep for a day" to the sweep
    tmp = "%s.%d.%s.tmp" % (p, os.getpid(), uuid.uuid4().hex[:8])
    try:
        with open(tmp, "w") as f:
            f.write("%s %d\n" % (socket.gethostname(), pid))
        os.replace(tmp, p)
    except OSError:
        try:
            os.unlink(tmp)
        except OSError:
            pass


def _pick_spool_base(out_dir: str, est_bytes: int) -> str:
    """/dev/shm when it is writable and has room for what the run may park there, else the output's folder.  With an
    unknown estimate (0) a tmpfs is only trusted when it has 1 GiB free - Docker's default 64 MB /dev/shm never is."""
    import os
    import shutil
    shm = "/dev/shm"
    if not (os.path.isdir(shm) and os.access(shm, os.W_OK)):
        return out_dir
    try:
        free = shutil.disk_usage(shm).free
    except OSError:
        return out_dir
    need = int(est_bytes * 1.15) + (64 << 20) if est_bytes > 0 else (1 << 30)
    floor = int(os.environ.get("PRISMA_SPOOL_MIN_FREE", "0"))
    return shm if free >= max(need, floor) else out_dir


def _sweep_stale_spools(base: str, min_age_s: float = 600.0, unknown_age_s: float = 86400.0):
    """Remove `prisma_spool.*` directories under `base` left by a crashed run.  The owner file names rank 0's host and pid: a
    directory whose owner is a dead process OF THIS HOST and in which nothing changed for ten minutes is nobody's.  A directory with no
    owner file, or one written on another host (output folders are often shared file systems, and `kill(pid, 0)` only speaks for this
    host's pid namespace), is presumed alive and only swept after a day without a change.  Age = the newest mtime of the
    directory and of the files in it - chunks waiting for rank 0 do not touch the directory entry."""
    import glob
    import os
    import shutil
    import socket
    import time
    host = socket.gethostname()
    for d in glob.glob(os.path.join(base, "prisma_spool.*")):
        try:
            if not os.path.isdir(d):
                continue
            newest = os.path.getmtime(d)
            for e in os.scandir(d):
                try:
                    newest = max(newest, e.stat().st_mtime)
                except OSError:
                    pass
            age = time.time() - newest
            if age < min_age_s:
                continue
            alive = unknown = False
            owners = glob.glob(os.path.join(d, "owner.*"))
            owners = [o for o in owners if not o.endswith(".tmp")]
            if not owners:
                unknown = True
            for o in owners:
                try:
                    txt = open(o).read().split()
                    pid = int(o.rsplit(".", 1)[1])
                    if len(txt) >= 2 and txt[0] != host:
                        unknown = True              # another host's run: cannot ask
                        continue
                    os.kill(pid, 0)
                    alive = True
                except PermissionError:             # somebody else's live process
                    alive = True
                except (OSError, ValueError):
                    pass
            if alive or (unknown and age < unknown_age_s):
                continue
            shutil.rmtree(d, ignore_errors=True)
        except OSError:
            pass


class Ranks:
    """One process per GPU under `torchrun` / `python -m torch.distributed.run` (RANK, LOCAL_RANK, WORLD_SIZE,
    MASTER_ADDR=127.0.0.1); a plain `python bands/<band>.py` is world 1 and touches neither torch nor a process group."""

    def __init__(self):
        import os
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            self.backend = os.environ.get("PRISMA_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
            if not dist.is_initialized():
                if self.backend == "nccl":
                    torch.cuda.set_device(self.device)
                # the only collectives are the scalar gather and short barriers, but they sit behind minutes of per-rank work of
                # different lengths (rank 0 also muxes): give the communicator watchdog hours, not torch's 10 minutes
                import datetime
                hours = float(os.environ.get("PRISMA_DIST_TIMEOUT_H", "12"))
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world, timeout=datetime.timedelta(hours=hours))

    @property
    def device(self) -> int:
        import os
        g = int(os.environ.get("PRISMA_GPUS_PER_NODE", "0"))       # set to 1 to let several ranks share one GPU (tests)
        return self.local_rank % g if g > 0 else self.local_rank

    @property
    def main(self) -> bool:
        return self.rank == 0

    def frames(self, n: int, halo: int = 0) -> Tuple[int, int]:
        return shard_range(n, self.rank, self.world, halo)

    def gather(self, local: np.ndarray, n: int, ctx=None):
        """Per-frame float32 scalars of every rank in frame order (rank 0 gets [n, k], the others None).  With a band ctx and
        the "nccl" backend the exchange is the library's own RCCL all-gather (pb_gather_scalars; the 128-byte communicator id
        travels through torch.distributed's object broadcast once); otherwise torch.distributed carries it (gloo in tests)."""
        import os
        if ctx is not None and self.world > 1 and self.backend == "nccl" and os.environ.get("PRISMA_NATIVE_GATHER", "1") != "0":
            import torch.distributed as dist
            if not getattr(ctx, "_comm_world", 0):
                box = [ctx.comm_unique_id() if self.main else None]
                dist.broadcast_object_list(box, src=0)
                ctx.comm_init(box[0], self.rank, self.world)
            local = np.ascontiguousarray(local, np.float32)
            local = local[:, None] if local.ndim == 1 else local
            per = -(-n // self.world)
            pad = np.zeros((per,) + local.shape[1:], np.float32)
            pad[: local.shape[0]] = local
            every = ctx.gather_scalars(pad)
            if not self.main:
                return None
            return np.concatenate([every[r][: shard_range(n, r, self.world)[1] - shard_range(n, r, self.world)[0]] for r in range(self.world)], 0)
        return gather_rows(local, n)

    def close(self):
        if self.world > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.barrier()
                dist.destroy_process_group()
