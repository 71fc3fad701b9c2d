"""Real checkpoint files through the band scripts (`load_weights` of every band handled `.pth` but nothing ever wrote
one).  The synthetic tensors are saved with torch in the wrappers the reference's checkpoints come in - DataParallel's `module.` prefix and
BatchNorm `num_batches_tracked` buffers for RAFT (bands/flow_raft.py:38-46), `{'meta', 'state_dict'}` for mmdet (init_detector),
`{'model': ...}` for GMFlow (bands/flow_gmflow.py:57-61) and ZoeDepth, a plain state dict for Depth-Anything, fp16 / bf16 storage - and read
back by the band's own loader (CPU), then the band CLI runs on the file and must write the bytes of the seeded-synthetic run (GPU)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bands"))

from prisma_amd import synth  # noqa: E402


def _tensors(w):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}


def _with_bn_counters(sd, marker=".running_var"):
    """every BatchNorm of a real checkpoint carries an int64 `num_batches_tracked` next to running_mean / running_var"""
    out = {}
    for k, v in sd.items():
        out[k] = v
        if k.endswith(marker) and k[: -len(marker)] + ".num_batches_tracked" not in sd:
            out[k[: -len(marker)] + ".num_batches_tracked"] = torch.tensor(12345, dtype=torch.int64)
    return out


def _same(got, ref, exact=True):
    floats = {k: v for k, v in got.items() if np.asarray(v).dtype.kind == "f"}
    ref = {k: v for k, v in ref.items() if np.asarray(v).dtype.kind == "f"}          # (the synthetic state dicts carry BatchNorm's integer counters too)
    assert set(floats) == set(ref), (sorted(set(floats) ^ set(ref))[:5])
    for k in ref:
        assert floats[k].dtype == np.float32 and floats[k].shape == ref[k].shape
        if exact:
            assert np.array_equal(floats[k], ref[k]), k
    return floats


def raft_ckpt(path):
    w = synth.raft_weights(seed=4321)
    sd = _with_bn_counters({"module." + k: v for k, v in _tensors(w).items()})
    torch.save(sd, path)
    return w, sd


def test_raft_checkpoint_prefix_and_bn_counters(tmp_path):
    import flow_raft as band
    p = str(tmp_path / "raft-sintel.pth")
    w, sd = raft_ckpt(p)
    n_int = sum(1 for v in sd.values() if not v.is_floating_point())
    assert n_int > 0, "the synthetic RAFT has BatchNorm layers in cnet: the checkpoint must carry their counters"
    got = band.load_weights(p)
    assert not any(k.startswith("module.") for k in got)
    _same(got, w)
    assert sum(1 for v in got.values() if np.asarray(v).dtype.kind == "i") == n_int        # kept as integers; the engine skips non-float entries


def test_half_and_bfloat16_checkpoints_become_float32(tmp_path):
    import flow_raft as band
    w = synth.raft_weights(seed=4321)
    for dt in (torch.float16, torch.bfloat16):
        p = str(tmp_path / ("raft_%s.pth" % str(dt).split(".")[-1]))
        torch.save({"module." + k: (v.to(dt) if v.is_floating_point() else v) for k, v in _tensors(w).items()}, p)
        got = _same(band.load_weights(p), w, exact=False)
        for k in list(w)[:8]:
            assert np.array_equal(got[k], torch.from_numpy(w[k]).to(dt).float().numpy()), (k, dt)


def test_mmdet_checkpoint_wrapper(tmp_path):
    import mask_mmdet as band
    cfg = synth.MASK_CFGS["tiny"]
    w = synth.solov2_weights(cfg)
    p = str(tmp_path / "solov2.pth")
    torch.save({"meta": {"mmdet_version": "2.25.0", "CLASSES": synth.COCO_CLASSES, "epoch": 36},
                "state_dict": _with_bn_counters(_tensors(w)), "optimizer": {"state": {}, "param_groups": []}}, p)
    got = band.load_weights(p, cfg)
    assert "meta" not in got and "optimizer" not in got
    _same(got, w)
    assert any(k.endswith("num_batches_tracked") for k in got)


def test_gmflow_zoe_and_depth_checkpoints(tmp_path):
    import depth_anything as dband
    import flow_gmflow as gband
    gw = synth.gmflow_weights(seed=2468)
    p = str(tmp_path / "gmflow_sintel.pth")
    torch.save({"model": _tensors(gw), "optimizer": {}, "step": 123}, p)
    _same(gband.load_weights(p), gw)
    torch.save(_tensors(gw), p)                                   # the bare dict is accepted too (reference :59-61)
    _same(gband.load_weights(p), gw)
    dw = synth.depth_anything_weights("vits", seed=1234)
    p = str(tmp_path / "depth_anything_vits14.pth")
    torch.save(_tensors(dw), p)
    _same(dband.load_weights("vits", p), dw)
    np.savez(str(tmp_path / "depth_anything_vits14.npz"), **dw)
    _same(dband.load_weights("vits", str(tmp_path / "depth_anything_vits14.npz")), dw)
    zw = synth.zoe_weights()
    p = str(tmp_path / "depth_anything_metric_depth_indoor.pt")
    torch.save({"model": _tensors(zw)}, p)
    _same(dband.load_metric_weights(p), zw)


def _clip(tmp_path, frames):
    folder = tmp_path / "clip"
    folder.mkdir()
    np.save(folder / "rgba.npy", frames)
    (folder / "metadata.json").write_text(json.dumps({"bands": {"rgba": {"url": "rgba.npy"}}}))
    os.environ["PRISMA_OVERWRITE"] = "1"
    return folder


@pytest.mark.gpu
def test_flow_raft_cli_on_a_pth_checkpoint(tmp_path, monkeypatch):
    import flow_raft as band
    folder = _clip(tmp_path, synth.frame_pair_sequence(3, 136, 168, seed=6))
    band.model = None
    band.main(["-i", str(folder), "--iterations", "3", "--scale", "1.0"])                 # seeded synthetic weights (PRISMA_SYNTH=1 in conftest)
    want, want_csv = np.load(folder / "flow_raft.npy").copy(), open(folder / "flow_raft.csv").read()
    band.model.close(); band.model = None
    p = str(tmp_path / "raft-sintel.pth")
    raft_ckpt(p)
    monkeypatch.setenv("PRISMA_SYNTH", "0")                                               # a missing checkpoint would now be an error
    # --small / --raft_model: parsed by the reference and without effect there (raft.py:28-53 builds the basic model regardless; init_model
    # loads args.model) - a reference command line carrying them runs unchanged
    band.main(["-i", str(folder), "--iterations", "3", "--scale", "1.0", "--model", p, "--raft_model", "models/raft-things.pth", "--small"])
    assert np.array_equal(np.load(folder / "flow_raft.npy"), want) and open(folder / "flow_raft.csv").read() == want_csv
    band.model.close(); band.model = None


@pytest.mark.gpu
def test_depth_mask_and_gmflow_cli_on_pth_checkpoints(tmp_path, monkeypatch):
    import depth_anything as dband
    import flow_gmflow as gband
    import mask_mmdet as mband
    folder = _clip(tmp_path, synth.frame_pair_sequence(3, 180, 300, seed=5))
    runs = [
        (dband, "depth_anything.npy", ["--encoder", "vits"], "--weights", lambda p: torch.save(_tensors(synth.depth_anything_weights("vits", seed=1234)), p)),
        (mband, "mask.npy", ["--arch", "tiny"], "--weights",
         lambda p: torch.save({"meta": {"CLASSES": synth.COCO_CLASSES}, "state_dict": _with_bn_counters(_tensors(synth.solov2_weights(synth.MASK_CFGS["tiny"])))}, p)),
        (gband, "flow_gmflow.npy", ["--scale", "1.0"], "--model", lambda p: torch.save({"model": _tensors(synth.gmflow_weights(seed=2468))}, p)),
    ]
    for band, out, extra, flag, write in runs:
        monkeypatch.setenv("PRISMA_SYNTH", "1")
        band.model = None
        band.main(["-i", str(folder)] + extra)
        want = np.load(folder / out).copy()
        band.model.close(); band.model = None
        p = str(tmp_path / (band.BAND + ".pth"))
        write(p)
        monkeypatch.setenv("PRISMA_SYNTH", "0")
        band.main(["-i", str(folder)] + extra + [flag, p])
        assert np.array_equal(np.load(folder / out), want), band.BAND
        band.model.close(); band.model = None
        if hasattr(band, "args"):
            band.args = None
