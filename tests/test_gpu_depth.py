"""GPU: the depth_anything band end to end through the C ABI vs the oracle and the committed
reference vectors.  Tolerance: BASELINE.json asks for 1e-3 relative on float depth; the
metric used here is max|d - ref| / max|ref| (error relative to the depth range, which is what
the min/max-normalised heat encoding sees) plus the relative L2 error.  Both are asserted < 1e-3 in the engine's
default precision (PB_PREC_SPLIT); the single-pass fp16 mode has its own documented bounds (conftest.TOL)."""
import os

import numpy as np
import pytest

from conftest import MARGIN_DEPTH_SPLIT, TOL, pw
from oracle import depth_oracle as O
from prisma_amd import engine, synth

pytestmark = pytest.mark.gpu

TOL_RANGE, TOL_L2 = TOL[1]      # max abs error / depth range, ||d - ref|| / ||ref||: 1e-3 each (split-fp16, the default)


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rell2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def report(tag, a, b):
    print(f"  {tag:14s} relmax {relmax(a, b):.3e}  relL2 {rell2(a, b):.3e}  {pw(a, b)}")


@pytest.mark.parametrize("prec", [1, 0])
@pytest.mark.parametrize("cfg,hw,seed", [("vits", (96, 128), 11), ("vitl_d4", (90, 120), 12)])
def test_small_models_stagewise(cfg, hw, seed, prec, golden_dir):
    c = synth.DEPTH_CFGS[cfg]
    w = synth.depth_anything_weights(c, seed=1234)
    frame = synth.frames(1, hw[0], hw[1], seed=seed)[0]
    net = engine.DepthAnything(w, c, device=0, max_batch=2, precision=prec)
    net.set_profiling(timing=False, debug_stages=True)
    depth, rgb, mn, mx = net.infer_batch(frame[None])
    x = O.preprocess(frame)[None]
    d_net, st = O.model_forward(w, x, c.depth, c.heads, return_stages=True)
    print(f"\n[{cfg}, precision {prec}] stage errors vs oracle")
    worst = 0.0
    for name, key in [("tokens", "tokens"), ("block0", "block0"), (f"block{c.depth - 1}", f"block{c.depth - 1}"),
                      ("feat0", "feat0"), ("feat3", "feat3"), ("layer1_rn", "layer1_rn"), ("layer4_rn", "layer4_rn"),
                      ("path4", "path4"), ("path3", "path3"), ("path2", "path2"), ("output_conv1", "output_conv1"),
                      ("net_depth", None)]:
        got = net.stage(name)
        ref = d_net if key is None else st[key]
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        report(name, got, ref)
        worst = max(worst, relmax(got, ref))
    ref = O.infer(w, frame, c.depth, c.heads)
    report("depth", depth[0], ref)
    z = np.load(os.path.join(golden_dir, f"depth_{cfg}_{hw[0]}x{hw[1]}.npz"))
    report("depth/golden", depth[0], z["depth"])
    assert relmax(depth[0], z["depth"]) < TOL[prec][0] and rell2(depth[0], z["depth"]) < TOL[prec][1]
    assert worst < 1e-2
    assert abs(mn[0] - depth[0].min()) == 0 and abs(mx[0] - depth[0].max()) == 0
    ref_rgb, _, _ = O.encode_depth_video(depth[0], flip=True)
    assert np.array_equal(rgb[0], ref_rgb)          # the fused encode is bit exact on the engine's own depth
    net.close()


def test_batch_matches_single():
    c = synth.DEPTH_CFGS["vits"]
    w = synth.depth_anything_weights(c, seed=1234)
    frames = synth.frames(3, 90, 160, seed=5)
    net = engine.DepthAnything(w, c, device=0, max_batch=2)     # 3 frames with max_batch 2 -> two chunks
    d3, rgb3, mn3, mx3 = net.infer_batch(frames)
    for i in range(3):
        d1, rgb1, mn1, mx1 = net.infer_batch(frames[i:i + 1])
        assert np.array_equal(d1[0], d3[i]) and np.array_equal(rgb1[0], rgb3[i]) and mn1[0] == mn3[i]
    ref = O.infer(w, frames[2], c.depth, c.heads)
    assert relmax(d3[2], ref) < TOL_RANGE
    net.close()


@pytest.mark.parametrize("prec", [1, 0])
def test_vitl_720p_against_reference_vectors(golden_dir, prec):
    """BASELINE configs[1]: one 1280x720 frame, ViT-L, batch 1, against the reference's own output."""
    z = np.load(os.path.join(golden_dir, "depth_vitl_720p.npz"))
    c = synth.DEPTH_CFGS["vitl"]
    w = synth.depth_anything_weights(c, seed=1234)
    frame = synth.frames(1, 720, 1280, seed=int(z["frame_seed"]))[0]
    net = engine.DepthAnything(w, c, device=0, max_batch=1, precision=prec)
    depth, rgb, mn, mx = net.infer_batch(frame[None])
    print()
    report(f"vitl 720p p{prec}", depth[0][::8, ::8], z["depth_s8"])
    assert relmax(depth[0][::8, ::8], z["depth_s8"]) < TOL[prec][0]
    assert rell2(depth[0][::8, ::8], z["depth_s8"]) < TOL[prec][1]
    assert abs(mx[0] - z["minmax"][1]) < TOL[prec][0] * z["minmax"][1]
    # size-independent properties at full size: encode is a pure function of (depth, min, max)
    ref_rgb, lo, hi = O.encode_depth_video(depth[0], flip=True)
    assert np.array_equal(rgb[0], ref_rgb) and lo == mn[0] and hi == mx[0]
    net.close()


def test_latency_context_splits_k_and_stays_within_tolerance():
    """A context created with max_batch = 1 (one frame per call: BASELINE configs[1]) lends its GEMM launches a split-K workspace (engine.h sk_ws_):
    its bits differ from those of a max_batch > 1 context - both are equally far from the reference - and are the same from run to run; contexts with
    max_batch > 1 never split, so there a frame's result does not depend on the batch it arrives in (the test below)."""
    c = synth.DEPTH_CFGS["vitl"]
    w = synth.depth_anything_weights(c, seed=1234)
    frame = synth.frames(1, 720, 1280, seed=5)
    one = engine.DepthAnything(w, c, device=0, max_batch=1, precision=1)
    d1, rgb1, mn1, mx1 = one.infer_batch(frame)
    d1b, rgb1b, _, _ = one.infer_batch(frame)
    one.set_profiling(True)
    one.infer_batch(frame)
    names = [k["name"] for k in one.kernel_stats()]
    one.close()
    many = engine.DepthAnything(w, c, device=0, max_batch=2, precision=1)
    d2, rgb2, mn2, mx2 = many.infer_batch(frame)
    many.close()
    assert np.array_equal(d1, d1b) and np.array_equal(rgb1, rgb1b)
    print()
    report("vitl 720p max_batch 1 (split-K) vs max_batch 2", d1[0], d2[0])
    # measured 3.2e-4 / 2.1e-4: far more than fp32 summation order by itself - a last-bit difference in an fp32 sum moves fp16 roundings of the
    # next layer's operands, i.e. it re-draws part of the rounding noise that makes up each context's own 3.3e-4 (L2) against the reference
    assert relmax(d1[0], d2[0]) < 6e-4 and rell2(d1[0], d2[0]) < 4e-4, (relmax(d1[0], d2[0]), rell2(d1[0], d2[0]))
    assert not np.array_equal(d1, d2), "max_batch = 1 did not take the split-K path (or the split is bit-identical, which fp32 addition is not)"
    assert any("gemm_kernel<128, 128" in n for n in names), names


@pytest.mark.parametrize("prec", [1, 0])
def test_head_tail_at_low_resolution_equals_the_reference_order(prec, monkeypatch, golden_dir):
    """dpt.py:158-160: output_conv1 -> bilinear resize to the network size -> output_conv2 (3 x 3, ReLU, 1 x 1, ReLU).  The engine takes the 3 x 3
    convolution's nine tap products at output_conv1's LOW resolution (one 1 x 1 GEMM with 288 columns) and resizes, sums, activates and projects
    them in one pass (engine.h wz_, elementwise.hip dpt_tail_kernel: both maps are linear, so they commute - a third of the matrix work and of the
    bytes, and the resized map is never rounded).  PB_HEAD_TAIL=0 (read at pb_create) keeps the reference's order: the two agree to rounding and
    are equally far from the reference's own output."""
    z = np.load(os.path.join(golden_dir, "depth_vitl_720p.npz"))
    c = synth.DEPTH_CFGS["vitl"]
    w = synth.depth_anything_weights(c, seed=1234)
    frame = synth.frames(1, 720, 1280, seed=int(z["frame_seed"]))
    new = engine.DepthAnything(w, c, device=0, max_batch=2, precision=prec)
    new.set_profiling(True)
    d_new = new.infer_batch(frame)[0]
    names_new = [k["name"] for k in new.kernel_stats()]
    new.close()
    monkeypatch.setenv("PB_HEAD_TAIL", "0")
    old = engine.DepthAnything(w, c, device=0, max_batch=2, precision=prec)
    old.set_profiling(True)
    d_old = old.infer_batch(frame)[0]
    names_old = [k["name"] for k in old.kernel_stats()]
    old.close()
    assert any("gemm_kernel<256, 32" in n for n in names_old) and not any("gemm_kernel<256, 32" in n for n in names_new), (names_old, names_new)
    print()
    report("vitl 720p p%d low-resolution head tail vs reference order" % prec, d_new[0], d_old[0])
    report("vitl 720p p%d low-resolution head tail vs reference" % prec, d_new[0][::8, ::8], z["depth_s8"])
    report("vitl 720p p%d reference-order head tail vs reference" % prec, d_old[0][::8, ::8], z["depth_s8"])
    tol = 4e-4 if prec else 1.5e-3
    assert relmax(d_new[0], d_old[0]) < tol, relmax(d_new[0], d_old[0])
    for d in (d_new, d_old):
        assert relmax(d[0][::8, ::8], z["depth_s8"]) < TOL[prec][0] and rell2(d[0][::8, ::8], z["depth_s8"]) < TOL[prec][1]


@pytest.mark.parametrize("prec", [1, 0])
def test_vitl_batch_32_at_1080p_equals_single_frames(prec):
    """BASELINE configs[3]: ViT-L on a batch of 32 1920x1080 frames (one engine call, max_batch 32 - the bench's shape).  Frame i
    of the batch equals the same frame run alone, bit for bit: depth, encoded bytes, min and max; and the frame that has a
    reference golden-sized sibling stays within tolerance of the oracle."""
    c = synth.DEPTH_CFGS["vitl"]
    w = synth.depth_anything_weights(c, seed=1234)
    frames = synth.frames(32, 1080, 1920, seed=77)
    net = engine.DepthAnything(w, c, device=0, max_batch=32, precision=prec)
    d, rgb, mn, mx = net.infer_batch(frames)
    assert d.shape == (32, 1080, 1920) and np.isfinite(d).all() and (mx > mn).all()
    for i in (0, 13, 31):
        d1, rgb1, mn1, mx1 = net.infer_batch(frames[i:i + 1])
        assert np.array_equal(d1[0], d[i]) and np.array_equal(rgb1[0], rgb[i]) and mn1[0] == mn[i] and mx1[0] == mx[i], i
    net.close()
    # the frames against the oracle: all three in the split mode, where the margin to the 1e-3 is asserted too (conftest MARGIN_DEPTH_SPLIT)
    for i in ((0, 13, 31) if prec == 1 else (13,)):
        ref = O.infer(w, frames[i], c.depth, c.heads)
        report(f"b32[{i}] p{prec}", d[i], ref)
        assert relmax(d[i], ref) < TOL[prec][0] and rell2(d[i], ref) < TOL[prec][1], i
        if prec == 1:
            assert relmax(d[i], ref) < MARGIN_DEPTH_SPLIT, (i, relmax(d[i], ref))


def test_1080p_maps_to_same_network_and_runs():
    c = synth.DEPTH_CFGS["vits"]
    w = synth.depth_anything_weights(c, seed=1234)
    net = engine.DepthAnything(w, c, device=0, max_batch=1)
    f = synth.frames(1, 1080, 1920, seed=2)
    d, rgb, mn, mx = net.infer_batch(f)
    assert d.shape == (1, 1080, 1920) and np.isfinite(d).all() and mx[0] > mn[0]
    ref = O.infer(w, f[0], c.depth, c.heads)
    assert relmax(d[0], ref) < TOL_RANGE
    net.close()


def test_1080p_batch_encode_round_trip_and_errors():
    """Size-independent properties at BASELINE's full frame size: the heat encoding decodes back (viewer
    contract, encode.py:36-64) to the normalised, flipped depth; min/max are exact; bad calls fail loudly."""
    c = synth.DEPTH_CFGS["vits"]
    w = synth.depth_anything_weights(c, seed=1234)
    net = engine.DepthAnything(w, c, device=0, max_batch=3)
    f = synth.frames(3, 1080, 1920, seed=8)
    d, rgb, mn, mx = net.infer_batch(f)
    for i in range(3):
        assert mn[i] == d[i].min() and mx[i] == d[i].max()
        norm = (d[i] - mn[i]) / (mx[i] - mn[i])
        heat = O.rgb_to_heat(rgb[i])
        assert np.abs(heat - (1.0 - norm)).max() < 0.012          # 8-bit ramp: ~3 counts of 255 over 0.65 turn
    with pytest.raises(engine._lib.PrismaBandsError, match="bad arguments"):
        net.infer_batch(np.zeros((0, 8, 8, 3), np.uint8))
    net.close()
    bad = dict(w)
    bad.pop("pretrained.blocks.3.mlp.fc1.weight")
    with pytest.raises(engine._lib.PrismaBandsError, match="missing weight 'pretrained.blocks.3.mlp.fc1.weight'"):
        engine.DepthAnything(bad, c, device=0)


def test_odd_aspect_ratio_baseline_config1_size():
    """934x440 (the README / BASELINE configs[0] image size) -> network 518x1106, 37x79 patches."""
    c = synth.DEPTH_CFGS["vits"]
    w = synth.depth_anything_weights(c, seed=1234)
    net = engine.DepthAnything(w, c, device=0, max_batch=1)
    f = synth.frames(1, 440, 934, seed=12)
    assert engine.net_size(440, 934) == (518, 1106)
    d = net.infer(f[0])
    ref = O.infer(w, f[0], c.depth, c.heads)
    report("934x440", d, ref)
    assert relmax(d, ref) < TOL_RANGE and rell2(d, ref) < TOL_L2
    net.close()
