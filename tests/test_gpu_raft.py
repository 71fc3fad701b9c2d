"""GPU: the flow_raft band through the C ABI vs the oracle and the committed reference vectors.
Flow tolerance: BASELINE.json asks for 1e-3 relative; metric = max|f - ref| / max|ref| and relative L2, both < 1e-3 in the
default precision (PB_PREC_SPLIT); the single-pass fp16 mode is checked against its documented bounds (FAST_TOL)."""
import os

import numpy as np
import pytest

from conftest import TOL, pw
from oracle import raft_oracle as R
from prisma_amd import engine, synth

pytestmark = pytest.mark.gpu
TOL_RANGE, TOL_L2 = TOL[1]
FAST_TOL = (3e-3, 2e-3)       # PB_PREC_F16: 12 recurrent iterations on fp16 operands (measured 1.1-1.5e-3 / 0.8-1.1e-3)


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rell2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="module")
def net():
    n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=1)
    yield n
    n.close()


def test_fast_mode_against_reference_vectors(golden_dir):
    n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=0)
    for name in ("raft_125x157.npz", "raft_131x181.npz"):
        z = np.load(os.path.join(golden_dir, name))
        h, w = [int(v) for v in z["hw"]]
        fr = synth.frame_pair_sequence(2, h, w, seed=int(z["frame_seed"]))
        flow, _, _ = n.infer_sequence(fr, scale=1.0, iters=int(z["iters"]), backward=True)
        for k, got, ref in (("fwd", flow[0, 0], z["fwd"]), ("bwd", flow[0, 1], z["bwd"])):
            print("\n  fp16 %s %s relmax %.3e relL2 %.3e" % (name, k, relmax(got, ref), rell2(got, ref)))
            assert relmax(got, ref) < FAST_TOL[0] and rell2(got, ref) < FAST_TOL[1]
    n.close()


def test_pair_against_reference_vectors(net, golden_dir):
    z = np.load(os.path.join(golden_dir, "raft_125x157.npz"))
    h, w = [int(v) for v in z["hw"]]
    fr = synth.frame_pair_sequence(2, h, w, seed=int(z["frame_seed"]))
    net.set_profiling(timing=False, debug_stages=True)
    flow, rgb, mx = net.infer_sequence(fr, scale=1.0, iters=int(z["iters"]), backward=True)
    net.set_profiling(timing=False, debug_stages=False)
    assert flow.shape == (1, 2, h, w, 2)
    # intermediates against the pinned oracle's (tests/golden/raft_125x157.npz holds channel-strided samples from the run in which
    # make_golden.py asserted oracle == reference; flow_lo below is the reference's own tensor): feature map
    # (extractor.py BasicEncoder), initial hidden state (raft.py:112-115), the first 9x9x4 lookup (corr.py:29-50), the flow after one
    # update (update.py:122-136) and the 1/8-resolution flow after all of them
    h8, w8 = z["flow_lo"].shape[2:]
    gfm = net.stage("fmap")                                   # [frames, 256, h8, w8]; golden fmap1 = [fwd: frame 0, bwd: frame 1][:, ::4]
    net0 = net.stage("net0").reshape(2, h8, w8, 128).transpose(0, 3, 1, 2)
    corr0 = net.stage("corr0")                                # [2, 324, h8, w8]
    it0 = net.stage("flow_it0").reshape(2, h8, w8, 2).transpose(0, 3, 1, 2)
    for name, got, ref in (("fmap1", gfm[:, ::4], z["fmap1"]), ("net0", net0[:, ::4], z["net0"]), ("corr0", corr0[:, ::3], z["corr0"]),
                           ("flow_it0", it0, z["flow_it0"])):
        print("\n  %-9s/golden relmax %.3e relL2 %.3e" % (name, relmax(got, ref), rell2(got, ref)), end="")
        if name == "flow_it0":
            # not an output: the first of twelve increments (a fraction of the final flow's magnitude).  Bound it at 2e-3 of its own range
            # and at 1e-3 of the range of the quantity it accumulates into, the 1/8-resolution flow
            assert relmax(got, ref) < 2e-3 and np.abs(got - ref).max() < TOL_RANGE * np.abs(z["flow_lo"]).max(), name
        else:
            assert relmax(got, ref) < TOL_RANGE and rell2(got, ref) < TOL_L2, name
    # stages vs the oracle (same weights, same frames)
    import torch
    import torch.nn.functional as F
    a = torch.from_numpy(fr[0]).permute(2, 0, 1).float()[None]
    c = torch.from_numpy(fr[1]).permute(2, 0, 1).float()[None]
    pad = R.pad_amounts(h, w)
    i1 = F.pad(torch.cat([a, c], 0), pad, mode="replicate").numpy()
    i2 = F.pad(torch.cat([c, a], 0), pad, mode="replicate").numpy()
    lo, up, st = R.raft_forward(synth.raft_weights(seed=4321), i1, i2, int(z["iters"]), return_stages=True)
    fmap = net.stage("fmap")
    print("\n  fmap      relmax %.3e relL2 %.3e" % (relmax(fmap, st["fmap1"]), rell2(fmap, st["fmap1"])))
    flo = net.stage("flow_lo").reshape(2, lo.shape[2], lo.shape[3], 2).transpose(0, 3, 1, 2)
    print("  flow_lo   relmax %.3e relL2 %.3e" % (relmax(flo, lo), rell2(flo, lo)))
    assert relmax(fmap, st["fmap1"]) < TOL_RANGE and relmax(flo, lo) < TOL_RANGE and relmax(flo, z["flow_lo"]) < TOL_RANGE
    for name, got, ref in (("fwd", flow[0, 0], z["fwd"]), ("bwd", flow[0, 1], z["bwd"])):
        print("  %s/golden relmax %.3e relL2 %.3e  max|flow| %.2f  %s" % (name, relmax(got, ref), rell2(got, ref), np.abs(ref).max(), pw(got, ref)))
        assert relmax(got, ref) < TOL_RANGE and rell2(got, ref) < TOL_L2
    # encode: max displacement and colours are functions of the engine's own flow
    ref_rgb, ref_mx = R.process_flow(flow[0, 0], exact_atan2=True)
    assert mx[0, 0] == ref_mx
    assert np.array_equal(rgb[0, 0], ref_rgb)          # byte exact (correctly rounded float32 arctan2 on both sides)
    host_rgb, _ = R.process_flow(flow[0, 0])           # numpy's own float32 arctan2 (SVML or libm, depends on this host's CPU)
    d = np.abs(rgb[0, 0].astype(int) - host_rgb.astype(int))
    print("  bytes differing from this host's np.arctan2(float32) path: %.2e (max |diff| %d)" % ((d > 0).mean(), d.max()))
    assert d.max() <= 1 and (d > 0).mean() < 2e-3


def test_odd_feature_grid_against_reference_vectors(net, golden_dir):
    """131x181 pads to 136x184 -> 17 x 23 feature pixels: the level-0 volume rows are padded to a multiple of 8."""
    z = np.load(os.path.join(golden_dir, "raft_131x181.npz"))
    h, w = [int(v) for v in z["hw"]]
    fr = synth.frame_pair_sequence(2, h, w, seed=int(z["frame_seed"]))
    flow, rgb, mx = net.infer_sequence(fr, scale=1.0, iters=int(z["iters"]), backward=True)
    for name, got, ref in (("fwd", flow[0, 0], z["fwd"]), ("bwd", flow[0, 1], z["bwd"])):
        print("\n  %s/golden relmax %.3e relL2 %.3e  max|flow| %.2f" % (name, relmax(got, ref), rell2(got, ref), np.abs(ref).max()))
        assert relmax(got, ref) < TOL_RANGE and rell2(got, ref) < TOL_L2


@pytest.mark.parametrize("prec", [1, 0])
def test_720p_batch_of_8_pairs_against_reference_vectors(golden_dir, prec):
    """BASELINE configs[2]: 8 consecutive pairs of 1280x720 frames, 12 GRU iterations, no --scale, one engine call
    (14 400^2 correlation volumes, 8-pair batching, 32-bit-offset guards) against the REAL reference's output on the same
    frames (tests/golden/raft_full.npz: 1/8-strided samples + float64 sums per pair)."""
    z = np.load(os.path.join(golden_dir, "raft_full.npz"))
    fr = synth.frame_pair_sequence(9, 720, 1280, seed=int(z["frame_seed"]))
    n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=prec)
    flow, rgb, mx = n.infer_sequence(fr, scale=1.0, iters=int(z["iters"]), backward=False)
    n.close()
    assert flow.shape == (8, 1, 720, 1280, 2)
    tol = TOL[1] if prec == 1 else FAST_TOL
    for i in range(8):
        got, ref = flow[i, 0][::8, ::8], z["fwd720_s8"][i]
        print("\n  p%d pair %d relmax %.3e relL2 %.3e  %s" % (prec, i, relmax(got, ref), rell2(got, ref), pw(got, ref)), end="")      # (1/8-strided samples of the reference's flow)
        assert relmax(got, ref) < tol[0] and rell2(got, ref) < tol[1], i
        f64 = flow[i, 0].astype(np.float64)
        sums = np.array([f64[..., 0].sum(), f64[..., 1].sum(), np.abs(f64).sum()])
        assert np.all(np.abs(sums - z["sums720"][i]) < tol[1] * z["sums720"][i][2])       # whole-frame sums (every pixel counted)
        ref_rgb, ref_mx = R.process_flow(flow[i, 0], exact_atan2=True)
        assert mx[i, 0] == ref_mx and np.array_equal(rgb[i, 0], ref_rgb)                  # encode byte exact at full size


def test_update_block_two_fp16_passes_switch_stays_a_working_path(golden_dir, monkeypatch):
    """PB_MX_UPD=0 (raft_engine.hip load()) runs the update block's residual pass as a second fp16 pass instead of on e4m3 copies of its
    maps (the default since round 4): the A/B switch must keep meeting the split mode's bound."""
    z = np.load(os.path.join(golden_dir, "raft_full.npz"))
    fr = synth.frame_pair_sequence(9, 720, 1280, seed=int(z["frame_seed"]))[:3]
    monkeypatch.setenv("PB_MX_UPD", "0")
    n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=1)
    flow, _, _ = n.infer_sequence(fr, scale=1.0, iters=int(z["iters"]), backward=False)
    n.close()
    for i in range(2):
        got, ref = flow[i, 0][::8, ::8], z["fwd720_s8"][i]
        print("\n  PB_MX_UPD=0 pair %d relmax %.3e relL2 %.3e" % (i, relmax(got, ref), rell2(got, ref)), end="")
        assert relmax(got, ref) < TOL[1][0] and rell2(got, ref) < TOL[1][1]


@pytest.mark.parametrize("prec", [1, 0])
def test_1080p_scaled_pair_against_reference_vectors(golden_dir, prec):
    """BASELINE configs[4] (flow part): a 1920x1080 pair at the band's default --scale 0.75 -> 810x1440 -> network 816x1440
    (18 360^2 volume), 12 iterations, against the reference network fed the same 8-bit cubic resize."""
    z = np.load(os.path.join(golden_dir, "raft_full.npz"))
    big = synth.frame_pair_sequence(2, 1080, 1920, seed=int(z["frame_seed"]) + 1)
    n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=prec)
    flow, rgb, mx = n.infer_sequence(big, scale=0.75, iters=int(z["iters"]), backward=False)
    n.close()
    assert flow.shape == (1, 1, 810, 1440, 2)
    tol = TOL[1] if prec == 1 else FAST_TOL
    got, ref = flow[0, 0][::8, ::8], z["fwd1080_s8"]
    print("\n  p%d 1080p x0.75 relmax %.3e relL2 %.3e  %s" % (prec, relmax(got, ref), rell2(got, ref), pw(got, ref)))
    assert relmax(got, ref) < tol[0] and rell2(got, ref) < tol[1]
    f64 = flow[0, 0].astype(np.float64)
    sums = np.array([f64[..., 0].sum(), f64[..., 1].sum(), np.abs(f64).sum()])
    assert np.all(np.abs(sums - z["sums1080"]) < tol[1] * z["sums1080"][2])


def test_scaled_sequence_matches_oracle(net):
    fr = synth.frame_pair_sequence(3, 192, 256, seed=4)
    flow, rgb, mx = net.infer_sequence(fr, scale=0.75, iters=6, backward=False)
    assert flow.shape == (2, 1, 144, 192, 2)
    w = synth.raft_weights(seed=4321)
    for i in range(2):
        fwd, _ = R.infer_pair(w, fr[i], fr[i + 1], scale=0.75, iters=6)
        print("\n  pair %d relmax %.3e relL2 %.3e  %s" % (i, relmax(flow[i, 0], fwd), rell2(flow[i, 0], fwd), pw(flow[i, 0], fwd)))
        assert relmax(flow[i, 0], fwd) < TOL_RANGE and rell2(flow[i, 0], fwd) < TOL_L2


def test_identical_frames_give_finite_encode(net):
    fr = synth.frame_pair_sequence(1, 128, 160, seed=9)
    flow, rgb, mx = net.infer_sequence(np.concatenate([fr, fr]), scale=1.0, iters=2)
    assert np.isfinite(flow).all() and rgb.shape == (1, 1, 128, 160, 3)


@pytest.mark.gpu
def test_fwdbwd_mask_bit_exact():
    """pb_flow_fwdbwd_mask == oracle restatement of compute_fwdbwd_mask (bands/common/flow.py:19-40), bit for bit,
    on smooth near-inverse flows (mixed True / False), on noise, and with flows leaving the frame."""
    from oracle import raft_oracle as ro
    rng = np.random.default_rng(17)
    h, w = 67, 93
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    fwd = np.stack([3.0 * np.sin(yy / 9.0) + 1.3, 2.0 * np.cos(xx / 7.0) - 0.4], -1).astype(np.float32)
    bwd = (-fwd + rng.standard_normal((h, w, 2)).astype(np.float32) * 0.6).astype(np.float32)
    noise_f = (rng.standard_normal((h, w, 2)) * 4).astype(np.float32)
    noise_b = (rng.standard_normal((h, w, 2)) * 4).astype(np.float32)
    far_f = np.full((h, w, 2), 70.0, np.float32)
    flows = np.stack([np.stack([fwd, bwd]), np.stack([noise_f, noise_b]), np.stack([far_f, -far_f])])
    net = engine.FlowRaft(synth.raft_weights(seed=4321))
    got = net.fwdbwd_mask(flows)
    net.close()
    assert got.shape == (3, 2, h, w) and got.dtype == np.bool_
    for i in range(3):
        mf, mb = ro.compute_fwdbwd_mask(flows[i, 0], flows[i, 1])
        assert np.array_equal(got[i, 0], mf) and np.array_equal(got[i, 1], mb)
    assert 0.2 < got[0, 0].mean() < 0.95          # the smooth case really mixes both outcomes


@pytest.mark.gpu
def test_sequence_masks_match_flows():
    """The fused path (pb_flow_infer_sequence_masks) returns the same flows as the plain backward call and masks that
    equal the oracle's on those flows."""
    from oracle import raft_oracle as ro
    frames = synth.frame_pair_sequence(3, 136, 184, seed=9)
    net = engine.FlowRaft(synth.raft_weights(seed=4321))
    flow, rgb, mx, mask = net.infer_sequence_masks(frames, scale=1.0, iters=4)
    flow2, rgb2, mx2 = net.infer_sequence(frames, scale=1.0, iters=4, backward=True)
    net.close()
    assert np.array_equal(flow, flow2) and np.array_equal(rgb, rgb2) and np.array_equal(mx, mx2)
    for i in range(2):
        mf, mb = ro.compute_fwdbwd_mask(flow[i, 0], flow[i, 1])
        assert np.array_equal(mask[i, 0], mf) and np.array_equal(mask[i, 1], mb)


def test_volume_kernel_is_bit_identical_to_the_generic_gemm(tmp_path):
    """volume.hip (A-stationary, persistent along the targets) accumulates over K in the order of the generic GEMM kernels: the whole
    flow must not change by a bit when the correlation volume goes through those instead (PB_VOLUME=0, read once per process).
    131x181 pads to a 17 x 23 grid: 391 source rows (not a multiple of 128) and level strides of 576 / 128 / 64 / 64 columns."""
    import subprocess
    import sys
    fr = synth.frame_pair_sequence(3, 131, 181, seed=33)
    n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=1)
    flow, rgb, mx = n.infer_sequence(fr, scale=1.0, iters=6, backward=True)
    n.close()
    out = str(tmp_path / "generic.npy")
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from prisma_amd import engine, synth; "
            "fr = synth.frame_pair_sequence(3, 131, 181, seed=33); n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=1); "
            "f, _, _ = n.infer_sequence(fr, scale=1.0, iters=6, backward=True); np.save(%r, f)" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), out))
    env = dict(os.environ, PB_VOLUME="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    assert np.array_equal(np.load(out), flow)


def test_flat_addressed_gemm_builds_give_the_same_bytes(tmp_path):
    """Every GEMM / convolution launch whose operands fit a buffer resource stages through the buffer path (`buffer_load ... lds`, out-of-range taps
    read zeros); operands beyond 4 GB take the flat-addressed build of the same kernel (padding taps read a zero page).  PB_GEMM_BUFFER=0 (read once
    per process) sends every launch down the flat builds - the 128 x 96 tile's chunk walk included: same K order, same bytes."""
    import subprocess
    import sys
    fr = synth.frame_pair_sequence(3, 131, 181, seed=35)
    n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=1)
    flow, rgb, mx = n.infer_sequence(fr, scale=1.0, iters=3, backward=True)
    n.close()
    out = str(tmp_path / "flat.npy")
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from prisma_amd import engine, synth; "
            "fr = synth.frame_pair_sequence(3, 131, 181, seed=35); n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=1); "
            "f, _, _ = n.infer_sequence(fr, scale=1.0, iters=3, backward=True); np.save(%r, f)" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), out))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PB_GEMM_BUFFER="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    assert np.array_equal(np.load(out), flow)


@pytest.mark.parametrize("prec", [1, 0])
def test_encoder_96_wide_tile_does_not_change_a_bit(prec):
    """Stage 2 of both encoders (64 -> 96 and 96 -> 96 convolutions, 96 carried as 128 channels in the maps) runs on the 128 x 96 tile
    (gemm.h TILE_128x96: a quarter fewer MFMAs than the 128-wide tile spends on padding columns).  With the K axis of the 128 x 128 tile
    ("tile_n96" = 1) it walks K in the same order and its epilogues do the same arithmetic: the flow of a ragged three-frame clip is the byte
    string the 128 x 128 tile gives ("tile_n96" = 0), in split-fp16 (e4m3 residual maps, instance norm in fnet, folded BatchNorm + skip adds
    in cnet) and in the single-pass mode.  The default ("tile_n96" = 2) also walks a packed-channel K axis on the 96 -> 96 convolutions of
    the split mode (conv_walk.h conv_cw3_word: only the 96 real channels of every tap, 28 K tiles instead of 36): another summation order,
    so equal to what a K order is worth in this precision scheme, not to the bit - measured 3.3e-4 of the range here, where the slice-major
    order of the SAME tiles (PB_TAPIN=2) is 1.8e-4 L2 from the tap-major one on the 125 x 157 vector and the packed-channel one 1.8e-4
    (tools/diag_cw3b.py; all three sit 2.2-2.4e-4 L2 from the reference) - and the single-pass mode, which has no such copy of the weights,
    stays equal to the bit."""
    fr = synth.frame_pair_sequence(3, 131, 181, seed=34)
    n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=prec)
    try:
        out = {}
        for mode in (2, 1, 0):
            n.set_option("tile_n96", mode)
            out[mode] = [np.asarray(a) for a in n.infer_sequence(fr, scale=1.0, iters=4, backward=True)]
    finally:
        n.set_option("tile_n96", 2)
        n.close()
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a, b)
    e = relmax(out[2][0], out[1][0])
    print("\n  packed-channel K axis against the per-tap one, precision %d: relmax %.2e" % (prec, e), end="")
    if prec == 0:
        assert np.array_equal(out[2][0], out[1][0])
    else:
        assert 0 < e < 6e-4, e          # (0 would mean the packed-channel path did not run)


@pytest.mark.parametrize("pinned", [False, True])
def test_host_pipeline_chunks_equal_one_call(net, pinned):
    """pb_flow_infer_sequence is a three-stage pipeline over chunks of frame pairs (H2D of chunk i + 1, the band on chunk i, D2H of chunk
    i - 1; a chunk re-encodes its halo frame): flows, encodes and maximum displacements must be those of one whole-sequence call, bit for
    bit, whatever the chunk size, from pageable arrays (pinned staging inside the library) and from page-locked ones (used directly)."""
    import torch
    fr = synth.frame_pair_sequence(8, 136, 168, seed=21)
    net.set_option("host_chunk", 64)
    flow1, rgb1, mx1 = net.infer_sequence(fr, scale=1.0, iters=3, backward=True)
    try:
        for chunk in (2, 3):
            net.set_option("host_chunk", chunk)
            if pinned:
                hf = torch.from_numpy(fr).pin_memory()
                of = torch.empty(flow1.shape, dtype=torch.float32).pin_memory()
                og = torch.empty(rgb1.shape, dtype=torch.uint8).pin_memory()
                flow, rgb, mx = net.infer_sequence(hf.numpy(), scale=1.0, iters=3, backward=True, out_flow=of.numpy(), out_rgb=og.numpy())
                assert flow.ctypes.data == of.data_ptr() and rgb.ctypes.data == og.data_ptr()
            else:
                flow, rgb, mx = net.infer_sequence(fr, scale=1.0, iters=3, backward=True)
            assert np.array_equal(flow, flow1) and np.array_equal(rgb, rgb1) and np.array_equal(mx, mx1), (chunk, pinned)
    finally:
        net.set_option("host_chunk", 0)


@pytest.mark.parametrize("pinned", [False, True])
def test_masks_host_pipeline_chunks_equal_one_call(net, pinned):
    """pb_flow_infer_sequence_masks runs the same chunked three-stage pipeline as pb_flow_infer_sequence (rounds 1-5: malloc, copy, run, copy,
    free per call; reference loop bands/flow_raft.py:63-64,103-113): flows, encodes, maximum displacements and the forward / backward
    consistency masks of every chunk size equal those of one whole-sequence call and of the device-pointer entry point, bit for bit."""
    import torch
    fr = synth.frame_pair_sequence(7, 136, 168, seed=23)
    net.set_option("host_chunk", 64)
    flow1, rgb1, mx1, mask1 = net.infer_sequence_masks(fr, scale=1.0, iters=3)
    assert mask1.any() and not mask1.all()
    try:
        for chunk in (2, 4):
            net.set_option("host_chunk", chunk)
            if pinned:
                hf = torch.from_numpy(fr).pin_memory()
                of = torch.empty(flow1.shape, dtype=torch.float32).pin_memory()
                og = torch.empty(rgb1.shape, dtype=torch.uint8).pin_memory()
                ok = torch.empty(mask1.shape, dtype=torch.uint8).pin_memory()
                flow, rgb, mx, mask = net.infer_sequence_masks(hf.numpy(), scale=1.0, iters=3, out_flow=of.numpy(), out_rgb=og.numpy(), out_mask=ok.numpy())
                assert flow.ctypes.data == of.data_ptr() and mask.ctypes.data == ok.data_ptr()
            else:
                flow, rgb, mx, mask = net.infer_sequence_masks(fr, scale=1.0, iters=3)
            assert np.array_equal(flow, flow1) and np.array_equal(rgb, rgb1) and np.array_equal(mx, mx1) and np.array_equal(mask, mask1), (chunk, pinned)
            _, _, mx2, mask2 = net.infer_sequence_masks(fr, scale=1.0, iters=3, want_flow=False, want_rgb=False)      # masks alone
            assert np.array_equal(mask2, mask1) and np.array_equal(mx2, mx1)
    finally:
        net.set_option("host_chunk", 0)


def test_pipeline_error_exit_leaves_no_copy_in_flight(net):
    """with page-locked caller buffers the copy engines write CALLER memory asynchronously; an error in chunk i (here: a bad
    iteration count, refused by the engine after chunk 0's frames are already on their way) must not return while chunk i - 1's D2H or chunk
    i's H2D is still running.  After the failed call the caller's result buffer is scribbled and must stay as scribbled (a late DMA would
    overwrite it), and the ctx must still produce the right bytes."""
    import time
    import torch
    fr = synth.frame_pair_sequence(6, 136, 168, seed=29)
    flow0, rgb0, mx0 = net.infer_sequence(fr, scale=1.0, iters=2)
    hf = torch.from_numpy(fr).pin_memory()
    of = torch.empty(flow0.shape, dtype=torch.float32).pin_memory()
    og = torch.empty(rgb0.shape, dtype=torch.uint8).pin_memory()
    net.set_option("host_chunk", 2)
    try:
        with pytest.raises(engine._lib.PrismaBandsError):
            net.infer_sequence(hf.numpy(), scale=1.0, iters=0, out_flow=of.numpy(), out_rgb=og.numpy())
        of.fill_(123.0); og.fill_(77)
        time.sleep(0.2)
        assert bool((of == 123.0).all()) and bool((og == 77).all())
        with pytest.raises(engine._lib.PrismaBandsError):       # an undersized frame: refused by the engine's geometry check inside chunk 0
            net.infer_sequence(np.zeros((3, 40, 40, 3), np.uint8), scale=1.0, iters=2)
        flow, rgb, mx = net.infer_sequence(hf.numpy(), scale=1.0, iters=2, out_flow=of.numpy(), out_rgb=og.numpy())
        assert np.array_equal(flow, flow0) and np.array_equal(rgb, rgb0) and np.array_equal(mx, mx0)
    finally:
        net.set_option("host_chunk", 0)
