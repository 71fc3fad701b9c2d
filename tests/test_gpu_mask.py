"""GPU: the mask_mmdet band (SOLOv2) through the C ABI vs oracle/solov2_oracle.py.

The oracle for this band is PARITY UNPINNED (mmcv / cv2 / the model config are absent from the build container, see
the oracle's header), so these tests show that the HIP path equals the restatement, not the reference itself.
Tolerances: every feature map 1e-3 of the stage's range (relative max) and 5e-4 relative L2, the bar of the other two bands (round 3:
PB_PREC_SPLIT splits this band's activations too); the preprocessing bytes and the Matrix-NMS survivors exact; the id image
against the fp32 oracle through `id_image_report` - equal, or every differing pixel counted and placed on the threshold it sits on."""
import numpy as np
import pytest
import torch

from oracle import solov2_oracle as SO
from prisma_amd import engine, synth

pytestmark = pytest.mark.gpu
TOL_RANGE, TOL_L2 = 1e-3, 5e-4
KEEP = [synth.COCO_CLASSES.index(c) for c in synth.BAND_CLASSES]


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rell2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="module")
def tiny():
    cfg = synth.MASK_CFGS["tiny"]
    w = synth.solov2_weights(cfg)
    net = engine.MaskMMDet(w, cfg, max_batch=2)
    net.set_profiling(True, True)
    yield cfg, w, net
    net.close()


def test_network_stages_match_oracle(tiny):
    cfg, w, net = tiny
    frames = synth.frames(2, 180, 300, seed=5)
    out = net.infer_batch(frames, 0.5, KEEP)
    assert out.shape == frames.shape
    nh, nw, Hp, Wp = engine.mask_net_size(cfg, 180, 300)
    xs, metas = zip(*[SO.preprocess(f, cfg) for f in frames])
    assert metas[0]["img_shape"] == (nh, nw) and metas[0]["pad_shape"] == (Hp, Wp)
    x = np.concatenate(xs)
    got_in = net.stage("input")
    assert np.array_equal(got_in, x), "fixed-point resize + normalise must be bit exact"
    kps, cps, mf, c, p = SO.network(w, cfg, x, return_feats=True)
    worst = 0.0
    for name, ref in ([(f"c{i + 2}", t) for i, t in enumerate(c)] + [(f"p{i + 2}", t) for i, t in enumerate(p)] +
                      [("mask_feats", mf)] + [(f"kernel_pred{i}", t) for i, t in enumerate(kps)] +
                      [(f"cls_logit{i}", t) for i, t in enumerate(cps)]):
        got = net.stage(name)
        ref = ref.numpy()
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        a, b = relmax(got, ref), rell2(got, ref)
        print("  %-14s relmax %.3e relL2 %.3e" % (name, a, b))
        worst = max(worst, a)
        assert a < TOL_RANGE and b < TOL_L2, name


def _oracle_post_from_engine(cfg, net, n, meta, b):
    """get_results of the oracle on the ENGINE's soft outputs (so the discrete outcomes must agree exactly)."""
    kps = [torch.from_numpy(net.stage(f"kernel_pred{l}")) for l in range(5)]
    cps = [torch.from_numpy(net.stage(f"cls_logit{l}")) for l in range(5)]
    mf = torch.from_numpy(net.stage("mask_feats"))
    if net.precision == engine._lib.PREC_F16:        # single fp16: the engine multiplies fp16 kernels with the fp16 mask features
        kps = [k.half().float() for k in kps]
    return SO.get_results(cfg, kps, cps, mf, meta["img_shape"], meta["ori_shape"], img_id=b, return_debug=True)


def id_image_report(tag, out_img, w, cfg, frame, confidence=0.5, soft_tol=2e-4):
    """The band's id image against the fp32 oracle end to end (north_star: bit-exact ids).  Equal -> returns 0.  Otherwise every differing
    pixel is counted and placed: with the same drawn instances on both sides a pixel can only differ at the last threshold
    (resized sigmoid > mask_thr, solov2_head.py:748-759), and its distance from that threshold in the ORACLE's soft mask must lie inside
    the soft outputs' tolerance; a different instance list is reported with the score that crossed 0.5 / --confidence and fails."""
    x, meta = SO.preprocess(frame, cfg)
    kps, cps, mf = SO.network(w, cfg, x)
    sc, lb, mk, dbg = SO.get_results(cfg, kps, cps, mf, meta["img_shape"], meta["ori_shape"], return_debug=True)
    ref = SO.band_mask(sc, lb, mk, synth.COCO_CLASSES, synth.BAND_CLASSES, confidence, meta["ori_shape"])
    diff = out_img[..., 0] != ref[..., 0]
    n = int(diff.sum())
    if n == 0:
        print("  %s: id image equals the fp32 oracle's (%d pixels, %d drawn)" % (tag, diff.size, int((ref[..., 0] != 0).sum())))
        return 0
    keep = np.array([synth.COCO_CLASSES[int(c)] in synth.BAND_CLASSES for c in lb.numpy()], bool)
    drawn = np.nonzero(keep & (sc.numpy() > 0.5) & (sc.numpy() > confidence))[0]
    assert len(drawn), "%s: %d pixels differ and the oracle draws nothing" % (tag, n)
    margin = np.abs(dbg["soft"][drawn].numpy() - cfg.mask_thr).min(0)          # distance of the nearest drawn instance from mask_thr
    at = margin[diff]
    on_thr = int((at < soft_tol).sum())
    near = np.sort(np.abs(sc.numpy()[keep] - 0.5))[:3]
    print("  %s: %d of %d pixels differ from the fp32 oracle; %d sit on the mask threshold (|sigmoid - %.2f| < %.0e in the oracle's resized "
          "soft mask, largest %.2e), %d elsewhere; instance scores nearest 0.5: %s" % (tag, n, diff.size, on_thr, cfg.mask_thr, soft_tol,
                                                                                      float(at.max()), n - on_thr, np.round(near, 4)))
    assert on_thr == n, "%s: %d differing pixels are not explained by the mask threshold (an instance-level decision flipped)" % (tag, n - on_thr)
    return n


def test_postprocess_exact_on_engine_outputs(tiny):
    cfg, w, net = tiny
    frames = synth.frames(2, 180, 300, seed=5)
    out = net.infer_batch(frames, 0.5, KEEP)
    for b in range(2):
        _, meta = SO.preprocess(frames[b], cfg)
        sc, lb, mk, dbg = _oracle_post_from_engine(cfg, net, 2, meta, b)
        g_sc, g_lb, g_mk, g_cand = net.instances(b, with_masks=True)
        print("  frame %d: %d candidates, %d instances, %d drawn" % (b, g_cand, len(g_sc), int((g_sc > 0.5).sum())))
        assert g_cand == dbg["n_candidates"] and len(g_sc) == len(sc) > 0
        assert np.array_equal(g_lb, lb.numpy())
        assert np.allclose(g_sc, sc.numpy(), rtol=2e-3, atol=1e-6)      # a pixel on the 0.5 edge may flip an area by one
        diff = int((g_mk != mk.numpy()).sum())
        print("  instance-mask pixels that differ: %d" % diff)
        # round 5 (tools/mask_final_stage.py, profiles/r05e_mask_final_stage.txt): the engine's final stage IS the oracle's final stage on the
        # same inputs - the id image (the drawn instances) bit for bit at this size and at 720p; what separates the id image from the
        # end-to-end fp32 oracle's is upstream of it (id_image_report)
        assert diff < 2e-4 * g_mk.size                                   # all ~100 instances, drawn or not (threshold pixels of their own)
        ref_img = SO.band_mask(sc, lb, mk, synth.COCO_CLASSES, synth.BAND_CLASSES, 0.5, meta["ori_shape"])
        assert np.array_equal(out[b], ref_img)
        assert out[b].max() > 0, "the synthetic detector must draw something"
        assert np.array_equal(out[b][..., 0], out[b][..., 1]) and np.array_equal(out[b][..., 0], out[b][..., 2])


def test_end_to_end_mask_image_close_to_oracle(tiny):
    cfg, w, net = tiny
    frames = synth.frames(1, 180, 300, seed=8)
    out = net.infer_batch(frames, 0.5, KEEP)[0]
    ref = SO.infer(w, cfg, frames[0], synth.COCO_CLASSES, synth.BAND_CLASSES, 0.5)
    mism = (out != ref).mean()
    print("  end-to-end mask image pixel mismatch vs fp32 oracle: %.3e (%d of %d pixels)" % (mism, int((out[..., 0] != ref[..., 0]).sum()), out[..., 0].size))
    if net.precision == engine._lib.PREC_F16:
        assert mism < 0.03          # discrete decisions on fp16-perturbed scores: instance boundaries and near-threshold cells
        return
    id_image_report("180x300 seed 8", out, w, cfg, frames[0])
    for seed in (5, 21):
        fr = synth.frames(2, 180, 300, seed=seed)
        o2 = net.infer_batch(fr, 0.5, KEEP)
        for b in range(2):
            id_image_report("180x300 seed %d frame %d" % (seed, b), o2[b], w, cfg, fr[b])


def test_mask_ids_do_not_depend_on_the_batch(tiny):
    """GroupNorm statistics are fixed-order sums over fixed 256-pixel chunks (no atomics): frame i of a batch of 3 equals the
    same frame alone, bit for bit, and a repeated call reproduces itself."""
    cfg, w, net = tiny
    frames = synth.frames(3, 180, 300, seed=21)
    out3 = net.infer_batch(frames, 0.5, KEEP)
    again = net.infer_batch(frames, 0.5, KEEP)
    assert np.array_equal(out3, again)
    for i in range(3):
        assert np.array_equal(net.infer_batch(frames[i:i + 1], 0.5, KEEP)[0], out3[i]), i


def test_keep_classes_and_confidence(tiny):
    cfg, w, net = tiny
    frames = synth.frames(1, 180, 300, seed=5)
    base = net.infer_batch(frames, 0.5, KEEP)[0]
    none = net.infer_batch(frames, 0.999, KEEP)[0]
    assert not none.any()
    sc, lb, _, _ = net.instances(0)
    allc = net.infer_batch(frames, 0.5, None)[0]
    assert (allc != 0).sum() >= (base != 0).sum()
    with pytest.raises(engine._lib.PrismaBandsError):
        net.infer_batch(frames, 0.5, [80])


def test_r101_720p_against_oracle():
    """BASELINE-size case: ResNet-101, one 1280x720 frame -> 1333x750 -> 768x1344 network input."""
    cfg = synth.MASK_CFGS["r101"]
    w = synth.solov2_weights(cfg)
    net = engine.MaskMMDet(w, cfg, max_batch=2)
    net.set_profiling(True, True)
    frames = synth.frames(2, 720, 1280, seed=2)
    out = net.infer_batch(frames, 0.5, KEEP)
    assert engine.mask_net_size(cfg, 720, 1280) == (750, 1333, 768, 1344)
    x, meta = SO.preprocess(frames[1], cfg)
    assert np.array_equal(net.stage("input")[1], x[0])
    kps, cps, mf, c, p = SO.network(w, cfg, x, return_feats=True)
    for name, ref in (("c5", c[3]), ("p2", p[0]), ("mask_feats", mf), ("kernel_pred0", kps[0]), ("cls_logit0", cps[0]),
                      ("cls_logit4", cps[4])):
        got = net.stage(name)[1:2]
        a, b = relmax(got, ref.numpy()), rell2(got, ref.numpy())
        print("  %-13s relmax %.3e relL2 %.3e" % (name, a, b))
        assert a < TOL_RANGE and b < TOL_L2, name
    id_image_report("720p frame 1", out[1], w, cfg, frames[1])
    sc, lb, mk, dbg = _oracle_post_from_engine(cfg, net, 2, meta, 1)
    g_sc, g_lb, g_mk, g_cand = net.instances(1, with_masks=True)
    print("  %d candidates, %d instances, %d over 0.5" % (g_cand, len(g_sc), int((g_sc > 0.5).sum())))
    assert g_cand == dbg["n_candidates"] and np.array_equal(g_lb, lb.numpy())
    assert np.allclose(g_sc, sc.numpy(), rtol=2e-3, atol=1e-6)      # a pixel on the 0.5 edge may flip an area by one
    ref_img = SO.band_mask(sc, lb, mk, synth.COCO_CLASSES, synth.BAND_CLASSES, 0.5, meta["ori_shape"])
    # the oracle's final stage on the engine's own soft outputs reproduces the engine's id image exactly (0 of 921 600 pixels, r05e)
    assert np.array_equal(out[1], ref_img) and out[1].any()
    st = {s["name"]: s for s in net.kernel_stats()}
    print("  kernel ms (2 frames):", {k: round(v["ms"], 2) for k, v in st.items()})
    net.close()


def test_r101_1080p_batch_against_oracle():
    """BASELINE configs[4] (mask part): ResNet-101 on 1920x1080 frames -> 1333x750 -> 768x1344, a batch of 3 with max_batch 2;
    the frame checked against the oracle sits in the second chunk, and its mask image equals the same frame run alone."""
    cfg = synth.MASK_CFGS["r101"]
    w = synth.solov2_weights(cfg)
    net = engine.MaskMMDet(w, cfg, max_batch=2)
    net.set_profiling(True, True)
    frames = synth.frames(3, 1080, 1920, seed=4)
    out = net.infer_batch(frames, 0.5, KEEP)
    assert engine.mask_net_size(cfg, 1080, 1920) == (750, 1333, 768, 1344)
    x, meta = SO.preprocess(frames[2], cfg)
    assert np.array_equal(net.stage("input")[0], x[0])                 # last chunk holds frame 2 alone
    kps, cps, mf = SO.network(w, cfg, x)
    for name, ref in (("mask_feats", mf), ("kernel_pred0", kps[0]), ("cls_logit0", cps[0]), ("cls_logit4", cps[4])):
        got = net.stage(name)[0:1]
        a, b = relmax(got, ref.numpy()), rell2(got, ref.numpy())
        print("  %-13s relmax %.3e relL2 %.3e" % (name, a, b))
        assert a < TOL_RANGE and b < TOL_L2, name
    id_image_report("1080p frame 2", out[2], w, cfg, frames[2])
    sc, lb, mk, dbg = _oracle_post_from_engine(cfg, net, 1, meta, 0)
    g_sc, g_lb, g_mk, g_cand = net.instances(2, with_masks=True)      # instance records are indexed by frame of the call
    assert g_cand == dbg["n_candidates"] and np.array_equal(g_lb, lb.numpy())
    assert (g_mk != mk.numpy()).mean() < 2e-4
    ref_img = SO.band_mask(sc, lb, mk, synth.COCO_CLASSES, synth.BAND_CLASSES, 0.5, meta["ori_shape"])
    assert (out[2] != ref_img).mean() < 5e-4 and out[2].any()
    assert np.array_equal(net.infer_batch(frames[2:3], 0.5, KEEP)[0], out[2])
    net.close()


def _blob_masks(n, H, W, seed):
    """id images like the band's: a few discs / rectangles / thin lines of 255 (and wrapped overlap counts) on black."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    out = np.zeros((n, H, W, 3), np.uint8)
    for f in range(n):
        acc = np.zeros((H, W), np.int64)
        for _ in range(int(rng.integers(1, 7))):
            cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(2, max(3, min(H, W) // 3))
            kind = rng.integers(0, 3)
            if kind == 0:
                acc += 255 * ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r)
            elif kind == 1:
                acc += 255 * ((abs(yy - cy) <= r // 2) & (abs(xx - cx) <= r))
            else:
                acc += 255 * (abs((yy - cy) * 3 - (xx - cx) * 2) <= 2)
        out[f] = (acc & 255).astype(np.uint8)[..., None]
    return out


@pytest.mark.parametrize("H,W,n", [(180, 300, 3), (720, 1280, 2), (1080, 1920, 2), (37, 53, 4)])
def test_sdf_green_matches_the_restatement(tiny, H, W, n):
    """the --sdf channel (reference mask_mmdet.py:64-69,150-152) is computed on the GPU - exact integer squared
    distances inside a +-64 window, byte looked up in the table made with the reference's float64 expression - and equals the host
    restatement (oracle band_sdf: scipy's exact EDT) byte for byte, including frames with no mask, no background, and one pixel."""
    _, _, net = tiny
    m = _blob_masks(n + 4, H, W, seed=H + n)
    m[n] = 0                                                 # all-empty frame
    m[n + 1] = 255                                           # all-full frame
    m[n + 2] = 0
    m[n + 2, H // 3, W // 2] = 255                           # one mask pixel
    m[n + 3] = 255
    m[n + 3, H - 1, 0] = 0                                   # one background pixel, in a corner
    net.set_sdf(True)
    try:
        got = net.sdf_green(m)
    finally:
        net.set_sdf(False)
    # degenerate frames saturate like snowy's INF-initialised transform: no corner gradient
    assert (got[n, ..., 1] == 0).all() and (got[n + 1, ..., 1] == 255).all()
    for f in range(len(m)):
        want = SO.band_sdf(m[f])
        bad = int((got[f] != want).sum())
        assert bad == 0, f"frame {f} of {H}x{W}: {bad} bytes differ (first at {np.argwhere(got[f] != want)[0]})"


def test_infer_batch_writes_the_sdf_when_asked(tiny):
    """set_sdf: the band's product path - the id image comes back with the field already in G (R and B untouched)."""
    cfg, w, net = tiny
    frames = synth.frames(2, 180, 300, seed=5)
    plain = net.infer_batch(frames, 0.5, KEEP)
    net.set_sdf(True)
    try:
        withsdf = net.infer_batch(frames, 0.5, KEEP)
    finally:
        net.set_sdf(False)
    assert np.array_equal(withsdf[..., 0], plain[..., 0]) and np.array_equal(withsdf[..., 2], plain[..., 2])
    for f in range(2):
        assert np.array_equal(withsdf[f], SO.band_sdf(plain[f]))
    assert np.array_equal(net.infer_batch(frames, 0.5, KEEP), plain)          # off again


@pytest.mark.parametrize("pinned", [False, True])
def test_host_pipeline_equals_device_path(tiny, pinned):
    """pb_mask_infer_batch is a pipeline over the engine's chunks of max_batch frames (every chunk's H2D up front, chunk i's id images -
    with the --sdf channel - back on a second copy stream while chunk i + 1 runs; reference loop bands/mask_mmdet.py:131-154): the bytes of
    the device-pointer entry point on the same frames, from pageable arrays (pinned staging on the ctx) and page-locked ones (used directly)."""
    import torch
    cfg, w, net = tiny                                        # max_batch 2: five frames = chunks of 2, 2, 1
    frames = synth.frames(5, 180, 300, seed=31)
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.zeros_like(d_in)
    for sdf in (False, True):
        net.set_sdf(sdf)
        try:
            net.infer_batch_dev(d_in.data_ptr(), 5, 180, 300, 0.5, KEEP, d_out.data_ptr())
            net.sync()
            ref = d_out.cpu().numpy()
            assert ref.any()
            if pinned:
                hf = torch.from_numpy(frames).pin_memory()
                ho = torch.zeros(frames.shape, dtype=torch.uint8).pin_memory()
                out = net.infer_batch(hf.numpy(), 0.5, KEEP, out=ho.numpy())
                assert out.ctypes.data == ho.data_ptr()
            else:
                out = net.infer_batch(frames, 0.5, KEEP)
            assert np.array_equal(out, ref), (sdf, pinned)
            assert len(net.instances(4)[0]) == len(net.instances(4)[1])          # per-frame results cover the whole call
        finally:
            net.set_sdf(False)
