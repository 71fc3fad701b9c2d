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
    jobs (or two bands of one job) never share a namespace and nobody deletes anybody else's files.  `base` is PRISMA_SPOOL if
    set, else /dev/shm when it exists (8 ranks x ~200 frames/s x 6.2 MB of 1080p frames is ~10 GB/s: memory, not the output
    filesystem), else the output's folder.  It holds, at most, the other ranks' encoded shards until rank 0 gets to them; with
    PRISMA_SPOOL_MAX_CHUNKS = K > 0 a producer waits while K of its chunks are unconsumed (bounded spool, but the producers then
    finish one after the other at rank 0's mux speed: the default 0 keeps every GPU computing).

    End of a run: the scalar all-gather happens BEFORE the drain - every rank reaches it as soon as its own compute
    is done - and the ranks > 0 then wait for rank 0's `done` FILE (wait_done), not inside a collective, so no communicator
    watchdog runs while rank 0 muxes (world - 1) / world of the video."""

    def __init__(self, ranks: "Ranks", out_path: str, timeout_s: float = 0.0):
        import os
        self.rk = ranks
        self.timeout = timeout_s or float(os.environ.get("PRISMA_RELAY_TIMEOUT_S", "21600"))
        self.max_chunks = int(os.environ.get("PRISMA_SPOOL_MAX_CHUNKS", "0"))
        self.dir = ""
        self._mine: List[str] = []
        if ranks.world > 1:
            import uuid
            import torch.distributed as dist
            base = os.environ.get("PRISMA_SPOOL") or ("/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK)
                                                      else os.path.dirname(os.path.abspath(out_path)))
            box = [os.path.join(base, "prisma_spool.%s.%s" % (os.path.basename(out_path), uuid.uuid4().hex[:12])) if ranks.main else None]
            dist.broadcast_object_list(box, src=0)
            self.dir = box[0]
            os.makedirs(self.dir, exist_ok=True)

    def _path(self, start: int) -> str:
        import os
        return os.path.join(self.dir, "chunk_%09d.npz" % start)

    def put(self, start: int, arrays: dict):
        """rank > 0: publish the chunk whose first unit (frame / pair index) is `start`."""
        import os
        import time
        if self.max_chunks > 0:
            t0 = time.time()
            while sum(os.path.exists(p) for p in self._mine) >= self.max_chunks:
                if time.time() - t0 > self.timeout:
                    raise TimeoutError(f"rank 0 did not consume rank {self.rk.rank}'s chunks within {self.timeout:.0f} s")
                time.sleep(0.005)
            self._mine = [p for p in self._mine if os.path.exists(p)]
        tmp = self._path(start) + ".tmp.%d" % self.rk.rank
        with open(tmp, "wb") as f:
            np.savez(f, **{k: np.ascontiguousarray(v) for k, v in arrays.items()})
        os.replace(tmp, self._path(start))
        self._mine.append(self._path(start))

    def drain(self, n_units: int, chunk: int, write):
        """rank 0: for every other rank, in rank (= frame) order, for every chunk start of its shard: wait for the file, call
        write(start, {name: array}), delete it."""
        import os
        import time
        for r in range(1, self.rk.world):
            first, last = shard_range(n_units, r, self.rk.world)
            for s in range(first, last, chunk):
                p, t0 = self._path(s), time.time()
                while not os.path.exists(p):
                    if time.time() - t0 > self.timeout:
                        raise TimeoutError(f"rank {r} did not deliver chunk {s} within {self.timeout:.0f} s")
                    time.sleep(0.005)
                with np.load(p) as z:
                    write(s, {k: z[k] for k in z.files})
                os.remove(p)

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
                time.sleep(0.02)
        import torch.distributed as dist
        dist.barrier()                       # short: every rank is past its file wait
        if self.rk.main:
            try:
                os.remove(done)
                os.rmdir(self.dir)
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
