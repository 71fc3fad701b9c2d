"""CPU: the RAFT oracle reproduces the committed reference vectors (tests/golden/raft_*.npz)."""
import os

import numpy as np

from oracle import raft_oracle as R
ro = R
from prisma_amd import synth


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


import pytest


@pytest.mark.parametrize("case", ["raft_125x157", "raft_131x181"])
def test_raft_pair_matches_reference(golden_dir, case):
    z = np.load(os.path.join(golden_dir, case + ".npz"))
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


def test_fwdbwd_mask_properties():
    """bands/common/flow.py:19-40 restatement: exactly-inverse integer flows are consistent wherever the warp stays
    inside the image; a constant offset of more than alpha_2 breaks every pixel; sub-pixel sampling is bilinear on
    the 1/32 grid."""
    h, w = 24, 40
    f = np.zeros((h, w, 2), np.float32); f[..., 0] = 3; f[..., 1] = -2
    m_f, m_b = ro.compute_fwdbwd_mask(f, -f)
    assert m_f.dtype == np.bool_ and m_f[2:, :w - 3].all() and not m_f[:, w - 3:].any() and not m_f[:2].any()
    assert m_b[:h - 2, 3:].all() and not m_b[:, :3].any()
    m_f, _ = ro.compute_fwdbwd_mask(f, -f + np.float32(1.5))
    assert not m_f.any()
    img = np.arange(h * w * 2, dtype=np.float32).reshape(h, w, 2)
    grid = np.zeros((h, w, 2), np.float32)
    grid[..., 0] = np.arange(w) + 0.25
    grid[..., 1] = np.arange(h)[:, None] + 0.5
    out = ro.remap_linear_const(img, grid)
    ref = 0.5 * (0.75 * img[:-1, :-1] + 0.25 * img[:-1, 1:]) + 0.5 * (0.75 * img[1:, :-1] + 0.25 * img[1:, 1:])
    assert np.allclose(out[:-1, :-1], ref, rtol=1e-6)
    assert np.allclose(out[-1, :-1], 0.5 * (0.75 * img[-1, :-1] + 0.25 * img[-1, 1:]), rtol=1e-6)   # bottom taps are border 0
    # 1/64 rounds half-to-even onto the 1/32 grid: 0.015625 * 32 = 0.5 -> 0
    grid[..., 0] = np.arange(w) + 1.0 / 64
    grid[..., 1] = np.arange(h)[:, None]
    assert np.array_equal(ro.remap_linear_const(img, grid), img)


def test_atan2_rn_is_the_correctly_rounded_float32_arctan2():
    """oracle atan2_rn (the arithmetic the HIP flow-encode kernel repeats operation for operation) against numpy: within 1e-15
    of the float64 / long double arctan2, and identical after rounding to float32 - on random inputs, on the axes, on signed
    zeros, and NaN in -> NaN out."""
    rng = np.random.default_rng(5)
    y = np.concatenate([rng.standard_normal(400000), rng.standard_normal(1000) * 1e-6, [0.0, 0.0, 1.0, -1.0, 0.0, 3.0, -3.0]]).astype(np.float32)
    x = np.concatenate([rng.standard_normal(400000), rng.standard_normal(1000), [1.0, -1.0, 0.0, 0.0, 0.0, 3.0, -3.0]]).astype(np.float32)
    mine = R.atan2_rn(y, x)
    ref = np.arctan2(y.astype(np.longdouble), x.astype(np.longdouble))
    assert float(np.abs(mine - ref).max()) < 1e-15
    assert np.array_equal(mine.astype(np.float32), ref.astype(np.float32))
    assert np.array_equal(mine.astype(np.float32), np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(np.float32))
    assert np.isnan(R.atan2_rn(np.array([np.nan, 1.0]), np.array([1.0, np.nan]))).all()
    # process_flow's two arctan2 paths differ only by numpy's own float32 arctan2 error: never more than one grey level
    f = (rng.standard_normal((64, 80, 2)) * 5).astype(np.float32)
    a, _ = R.process_flow(f)
    b, _ = R.process_flow(f, exact_atan2=True)
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1
