"""CPU: the RAFT oracle reproduces the committed reference vectors (tests/golden/raft_*.npz)."""
import os

import numpy as np

from oracle import raft_oracle as R
from prisma_amd import synth


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def test_raft_pair_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "raft_125x157.npz"))
    h, w = [int(v) for v in z["hw"]]
    fr = synth.frame_pair_sequence(2, h, w, seed=int(z["frame_seed"]))
    wts = synth.raft_weights(seed=4321)
    fwd, bwd = R.infer_pair(wts, fr[0], fr[1], scale=1.0, iters=int(z["iters"]))
    assert fwd.shape == (h, w, 2)
    assert rel(fwd, z["fwd"]) < 1e-4 and rel(bwd, z["bwd"]) < 1e-4
    rgb, mx = R.process_flow(z["fwd"])
    assert np.array_equal(rgb, z["fwd_rgb"]) and mx == z["fwd_max"]


def test_flow_encode_vectors(golden_dir):
    z = np.load(os.path.join(golden_dir, "encode.npz"))
    rgb, mx = R.process_flow(z["flow"])
    assert np.array_equal(rgb, z["flow_rgb"]) and mx == z["flow_max"]
    rgb0, m0 = R.process_flow(np.zeros((8, 8, 2), np.float32))     # last frame of every video: 0/0 -> NaN -> 0
    assert np.array_equal(rgb0, z["zero_rgb"]) and m0 == 0.0 and not rgb0.any()


def test_padder_and_resize_geometry():
    assert R.pad_amounts(540, 960) == [0, 0, 2, 2]                 # 720p * 0.75 -> 544 x 960 (SURVEY a-2.1)
    assert R.pad_amounts(810, 1440) == [0, 0, 3, 3]
    assert R.pad_amounts(720, 1280) == [0, 0, 0, 0]
    assert R.scaled_size(720, 1280, 0.75) == (540, 960) and R.scaled_size(1080, 1920, 0.75) == (810, 1440)
    img = np.random.default_rng(0).integers(0, 256, (40, 64, 3), dtype=np.uint8)
    out = R.cv_resize_cubic_u8(img, 0.75)
    assert out.shape == (30, 48, 3) and out.dtype == np.uint8
    flat = np.full((16, 16, 3), 77, np.uint8)
    assert (R.cv_resize_cubic_u8(flat, 0.75) == 77).all()          # coefficients sum to 2048 -> constants survive
