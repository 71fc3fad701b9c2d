"""CPU: the flow_gmflow oracle (oracle/gmflow_oracle.py) against the vectors the REAL reference produced (oracle/make_golden.py gmflow:
bands/gmflow/gmflow.py GMFlow + InputPadder(padding_factor=16), called like bands/flow_gmflow.py:66-118).  The HIP path of this band
is not built yet; this pins the checker it will be held against."""
import os

import numpy as np
import pytest

from oracle import gmflow_oracle as G
from prisma_amd import synth


def relmax(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("name,bidir", [("gmflow_125x157.npz", True), ("gmflow_216x300.npz", False)])
def test_oracle_reproduces_the_reference_flow(golden_dir, name, bidir):
    z = np.load(os.path.join(golden_dir, name))
    h, w = [int(v) for v in z["hw"]]
    fr = synth.frame_pair_sequence(2, h, w, seed=int(z["frame_seed"]))
    wts = synth.gmflow_weights(seed=2468)
    fwd, bwd = G.infer_pair(wts, fr[0], fr[1], scale=1.0, backward=bidir)
    assert fwd.shape == (h, w, 2) and relmax(fwd, z["fwd"]) < 1e-5
    if bidir:
        assert relmax(bwd, z["bwd"]) < 1e-5
    else:
        assert bwd is None


def test_stages_and_geometry(golden_dir):
    z = np.load(os.path.join(golden_dir, "gmflow_125x157.npz"))
    assert G.pad_amounts(125, 157) == [1, 2, 1, 2] and G.pad_amounts(810, 1440) == [0, 0, 3, 3] and G.pad_amounts(128, 160) == [0, 0, 0, 0]
    fr = synth.frame_pair_sequence(2, 125, 157, seed=int(z["frame_seed"]))
    ims = [np.pad(fr[i].transpose(2, 0, 1)[None].astype(np.float32), ((0, 0), (0, 0), (1, 2), (1, 2)), mode="edge") for i in (0, 1)]
    up, st = G.gmflow_forward(synth.gmflow_weights(seed=2468), ims[0], ims[1], bidir=True, return_stages=True)
    assert up.shape == (2, 2, 128, 160)
    for k, got in (("feat0", st["feat0"][:, ::4]), ("block0", st["block0"][:, :, ::4]), ("tfeat0", st["tfeat0"][:, ::4]),
                   ("flow_match", st["flow_match"]), ("flow_prop", st["flow_prop"])):
        assert relmax(got, z[k]) < 1e-5, k
    # the shifted-window mask separates the wrapped-around regions (transformer.py:18-44): 3 x 3 regions, window-local
    m = G.shift_mask(16, 20, 8, 10).numpy()
    assert m.shape == (4, 80, 80) and set(np.unique(m)) == {-100.0, 0.0} and (m[0] == 0).all() and (m[3] != 0).any()
    # position embedding: one window's sine table tiled over the 2 x 2 windows (utils.py:61-86)
    import torch
    f0, _ = G.add_position(torch.zeros(1, 128, 16, 20), torch.zeros(1, 128, 16, 20), 2)
    assert torch.equal(f0[0, :, :8, :10], f0[0, :, 8:, 10:]) and not torch.equal(f0[0, :, :8, :10], f0[0, :, :8, 1:11])


def test_weight_names_match_the_reference_state_dict():
    names = [n for n, _ in synth.gmflow_param_shapes()]
    assert len(names) == 123 and len(set(names)) == 123
    assert sum(int(np.prod(s)) for _, s in synth.gmflow_param_shapes()) == 4680288          # the published 4.7 M parameters
    assert "transformer.layers.5.cross_attn_ffn.mlp.2.weight" in names and "feature_flow_attn.k_proj.bias" in names
