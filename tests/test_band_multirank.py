"""GPU: the band scripts under `torch.distributed.run` with 2 ranks (sharing GPU 0, gloo for the gather) produce the
same files as a single process - SURVEY 8(e): contiguous frame shards, 1-frame halo for flow pairs, rank 0 muxes."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(script, args, ranks, nccl=False):
    env = dict(os.environ, PRISMA_OVERWRITE="1", PRISMA_BATCH="2")
    if nccl:        # one GPU per rank, torch.distributed "nccl" (= RCCL) and the native pb_comm / pb_gather_scalars path of shard.Ranks
        env.pop("PRISMA_DIST_BACKEND", None); env.pop("PRISMA_GPUS_PER_NODE", None)
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    else:
        env.update(PRISMA_DIST_BACKEND="gloo", PRISMA_GPUS_PER_NODE="1")
    cmd = [sys.executable]
    if ranks > 1:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bands", script)] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def _clip(tmp_path, name, frames):
    folder = tmp_path / name
    folder.mkdir()
    np.save(folder / "rgba.npy", frames)
    (folder / "metadata.json").write_text(json.dumps({"bands": {"rgba": {"url": "rgba.npy"}}}))
    return folder


@pytest.mark.parametrize("script,extra,outputs", [
    ("depth_anything.py", ["--encoder", "vits"], ["depth_anything.npy", "depth_anything_min.csv", "depth_anything_max.csv"]),
    ("flow_raft.py", ["--iterations", "3", "--scale", "1.0", "-b", "--mask"],
     ["flow_raft.npy", "flow_raft_bwd.npy", "flow_raft_mask.npy", "flow_raft_mask_bwd.npy", "flow_raft.csv"]),
    ("mask_mmdet.py", ["--arch", "tiny"], ["mask.npy"]),
    ("flow_gmflow.py", ["--scale", "1.0", "-b"], ["flow_gmflow.npy", "flow_gmflow_bwd.npy", "flow_gmflow.csv"]),
])
def test_two_ranks_equal_one(tmp_path, script, extra, outputs):
    from prisma_amd import synth
    frames = synth.frame_pair_sequence(5, 176, 256, seed=6) if script.startswith("flow") else synth.frames(5, 180, 300, seed=5)
    a, b = _clip(tmp_path, "one", frames), _clip(tmp_path, "two", frames)
    _run(script, ["-i", str(a)] + extra, 1)
    _run(script, ["-i", str(b)] + extra, 2)
    for name in outputs:
        if name.endswith(".npy"):
            x, y = np.load(a / name), np.load(b / name)
            assert x.shape == y.shape and x.shape[0] == 5
            # a frame's result does not depend on its chunk or rank: InstanceNorm / GroupNorm statistics are fixed-order
            # two-pass sums, every GEMM tile shape accumulates K in the same order, min / max atomics are exact
            assert np.array_equal(x, y), name
        else:
            assert open(a / name).read() == open(b / name).read()
    ma, mb = json.load(open(a / "metadata.json")), json.load(open(b / "metadata.json"))
    assert ma == mb


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("script,extra,outputs", [
    ("depth_anything.py", ["--encoder", "vits"], ["depth_anything.npy", "depth_anything_min.csv", "depth_anything_max.csv"]),
    ("flow_raft.py", ["--iterations", "3", "--scale", "1.0"], ["flow_raft.npy", "flow_raft.csv"]),
])
def test_two_gpus_over_rccl_equal_one(tmp_path, script, extra, outputs):
    """The same comparison with one GPU per rank on the nccl (= RCCL over xGMI) backend: the per-frame scalars travel through
    `shard.Ranks.gather` on device buffers.  Needs two GPUs - skipped on the single-GPU boxes the suite usually runs on."""
    if _gpus() < 2:
        pytest.skip("needs 2 GPUs (this box has %d)" % _gpus())
    from prisma_amd import synth
    frames = synth.frame_pair_sequence(5, 176, 256, seed=6) if script.startswith("flow") else synth.frames(5, 180, 300, seed=5)
    a, b = _clip(tmp_path, "one", frames), _clip(tmp_path, "two", frames)
    _run(script, ["-i", str(a)] + extra, 1)
    _run(script, ["-i", str(b)] + extra, 2, nccl=True)
    for name in outputs:
        if name.endswith(".npy"):
            assert np.array_equal(np.load(a / name), np.load(b / name)), name
        else:
            assert open(a / name).read() == open(b / name).read()
