"""GPU: parity on HEAVY-TAILED weights (VERDICT r2 item 5 / ADVICE r2): the split-precision mode stores e4m3 copies of activations at
fixed power-of-two scales, and every other parity test uses unit-scale synthetic weights whose activations never come near the copies'
saturation point.  prisma_amd/synth.py *_heavy re-scales outlier channels (LayerNorm / BatchNorm gains x 30-50, DPT conv channels x 30)
and tests/golden/{depth_vitl_heavy_720p,raft_heavy_184x256}.npz hold what the REAL reference models return on those weights
(oracle/make_golden.py heavy; activations reach 130+ in the DPT head, ~150 behind the LayerNorms).  The bound is the same 1e-3
(max / range and L2) as everywhere else.  The round-2 scales (2^4 on token copies, 2^3 / 2^15 on map copies: saturation at 28 / 56)
are run beside the shipped ones (2^0: 448) and their error printed, so the difference the scales make is on record."""
import os

import numpy as np
import pytest

from conftest import MARGIN_DEPTH_SPLIT, TOL
from prisma_amd import engine, synth

pytestmark = pytest.mark.gpu


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rell2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _with_scales(monkeypatch, a8, lo8):
    for k, v in (("PB_A8_POW", a8), ("PB_LO8_POW", lo8)):
        if v is None:
            monkeypatch.delenv(k, raising=False)
        else:
            monkeypatch.setenv(k, str(v))


def test_depth_vitl_heavy_tailed_weights(golden_dir, monkeypatch):
    z = np.load(os.path.join(golden_dir, "depth_vitl_heavy_720p.npz"))
    c = synth.DEPTH_CFGS["vitl"]
    w = synth.depth_anything_weights_heavy(c, seed=1234)
    frame = synth.frames(1, 720, 1280, seed=int(z["frame_seed"]))[0]
    res = {}
    for tag, a8, lo8 in (("shipped scales", None, None), ("round-2 scales 2^4 / 2^3", 4, 3)):
        _with_scales(monkeypatch, a8, lo8)          # read when the engine is constructed
        net = engine.DepthAnything(w, c, device=0, max_batch=1, precision=1)
        d = net.infer_batch(frame[None], want_rgb=False)[0][0]
        net.close()
        res[tag] = (relmax(d[::8, ::8], z["depth_s8"]), rell2(d[::8, ::8], z["depth_s8"]))
        print("\n  ViT-L heavy-tailed 720p, %s: relmax %.3e relL2 %.3e" % ((tag,) + res[tag]))
    _with_scales(monkeypatch, None, None)
    net = engine.DepthAnything(w, c, device=0, max_batch=1, precision=0)
    d = net.infer_batch(frame[None], want_rgb=False)[0][0]
    net.close()
    print("  ViT-L heavy-tailed 720p, single-pass fp16: relmax %.3e relL2 %.3e" % (relmax(d[::8, ::8], z["depth_s8"]), rell2(d[::8, ::8], z["depth_s8"])))
    e = res["shipped scales"]
    assert e[0] < TOL[1][0] and e[1] < TOL[1][1], res


def test_depth_vitl_heavy_tailed_weights_at_1080p_keep_the_margin(golden_dir):
    """the heavy-tailed weights at the bench's frame size (frame 13 of the 32 x 1080p clip) against what the REAL
    reference returns on them (tests/golden/depth_vitl_heavy_1080p.npz) - inside the 1e-3, and inside the margin the per-layer residual
    assignment is allowed to use (conftest MARGIN_DEPTH_SPLIT)."""
    z = np.load(os.path.join(golden_dir, "depth_vitl_heavy_1080p.npz"))
    c = synth.DEPTH_CFGS["vitl"]
    w = synth.depth_anything_weights_heavy(c, seed=1234)
    frame = synth.frames(32, 1080, 1920, seed=int(z["frame_seed"]))[int(z["frame_index"])]
    net = engine.DepthAnything(w, c, device=0, max_batch=1, precision=1)
    d = net.infer_batch(frame[None], want_rgb=False)[0][0]
    net.close()
    e = (relmax(d[::8, ::8], z["depth_s8"]), rell2(d[::8, ::8], z["depth_s8"]))
    print("\n  ViT-L heavy-tailed 1080p (frame 13 of the bench clip): relmax %.3e relL2 %.3e" % e)
    assert e[0] < MARGIN_DEPTH_SPLIT and e[1] < TOL[1][1], e
    s = d.astype(np.float64)
    assert abs(s.sum() - z["depth_sum"][0]) < TOL[1][1] * z["depth_sum"][1]


def test_raft_heavy_tailed_weights(golden_dir, monkeypatch):
    z = np.load(os.path.join(golden_dir, "raft_heavy_184x256.npz"))
    h, w = [int(v) for v in z["hw"]]
    fr = synth.frame_pair_sequence(2, h, w, seed=int(z["frame_seed"]))
    rw = synth.raft_weights_heavy(seed=4321)
    res = {}
    for tag, a8, lo8 in (("shipped scales", None, None), ("round-2 scales 2^4 / 2^3", 4, 3)):
        _with_scales(monkeypatch, a8, lo8)
        net = engine.FlowRaft(rw, device=0, precision=1)
        flow, _, _ = net.infer_sequence(fr, scale=1.0, iters=int(z["iters"]), backward=True, want_rgb=False)
        net.close()
        res[tag] = max(relmax(flow[0, 0], z["fwd"]), relmax(flow[0, 1], z["bwd"])), max(rell2(flow[0, 0], z["fwd"]), rell2(flow[0, 1], z["bwd"]))
        print("\n  RAFT heavy-tailed %dx%d, %s: relmax %.3e relL2 %.3e" % ((h, w, tag) + res[tag]))
    _with_scales(monkeypatch, None, None)
    e = res["shipped scales"]
    assert e[0] < TOL[1][0] and e[1] < TOL[1][1], res
