"""GPU: `depth_anything --metric` (ZoeDepth head over the ViT-L core) through the C ABI vs oracle/zoe_oracle.py.
Tolerance: north_star's 1e-3 relative on the band's output (metric depth, at network and at frame resolution), asserted in the
default precision (PB_PREC_SPLIT); intermediate stages (relative depth of the core, bin centres) are printed and bounded at 2e-3."""
import numpy as np
import pytest

from oracle import zoe_oracle as Z
from prisma_amd import engine, synth

pytestmark = pytest.mark.gpu
TOL_RANGE, TOL_L2 = 1e-3, 1e-3
STAGE_RANGE, STAGE_L2 = 2e-3, 1e-3


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rell2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def test_metric_depth_matches_oracle():
    w = synth.zoe_weights()
    net = engine.DepthAnything(w, "vitl", max_batch=2, metric=True)
    net.set_profiling(True, True)
    frames = synth.frames(2, 360, 640, seed=3)
    depth, rgb, mn, mx = net.infer_batch(frames, want_depth=True, want_rgb=True, flip=False)
    assert depth.shape == (2, 360, 640) and rgb.shape == (2, 360, 640, 3)
    ref_net, st = Z.forward(w, Z.preprocess(frames[1]), return_stages=True)
    for name, key in (("net_depth", "rel_depth"), ("seed_bins", "seed_bins"), ("bins0", "bins0"), ("bins3", "bins3"), ("metric_net", None)):
        got = net.stage(name)[1]
        ref = ref_net[0] if key is None else np.asarray(st[key])[0]
        if name.startswith("bins") or name == "seed_bins":
            ref = ref.reshape(64, -1).T                      # oracle NCHW -> engine [pixels, 64]
        a, b = relmax(got, ref), rell2(got, ref)
        print("  %-10s relmax %.3e relL2 %.3e" % (name, a, b))
        if name == "metric_net":
            assert a < TOL_RANGE and b < TOL_L2, name
        else:
            assert a < STAGE_RANGE and b < STAGE_L2, name
    # the Pillow resize of the engine's own network output is exact (double accumulation, float32 passes)
    from PIL import Image
    pil = np.asarray(Image.fromarray(net.stage("metric_net")[1]).resize((640, 360)))
    assert np.array_equal(depth[1], pil)
    ref = Z.pil_resize_f32(ref_net[0], 360, 640)
    print("  final      relmax %.3e relL2 %.3e  range %.2f .. %.2f m" % (relmax(depth[1], ref), rell2(depth[1], ref), ref.min(), ref.max()))
    assert relmax(depth[1], ref) < TOL_RANGE and rell2(depth[1], ref) < TOL_L2
    assert abs(mn[1] - depth[1].min()) < 1e-6 and abs(mx[1] - depth[1].max()) < 1e-6
    # not flipped: the farthest pixel is hot (heat 1 -> red), bands/depth_anything.py:150,188
    far = np.unravel_index(depth[1].argmax(), depth[1].shape)
    assert tuple(rgb[1][far]) == (255, 0, 0)
    net.close()
