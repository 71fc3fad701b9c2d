"""CPU: the ZoeDepth metric-head oracle (oracle/zoe_oracle.py) against the reference's own layer modules (vectors made
by oracle/make_golden.py zoe, which loads patchfusion/zoedepth/models/layers/*.py) and against Pillow itself."""
import os

import numpy as np
import torch

from oracle import zoe_oracle as Z
from prisma_amd import synth


def test_head_layers_match_reference_vectors(golden_dir):
    z = np.load(os.path.join(golden_dir, "zoe_layers.npz"))
    w = {k: v for k, v in synth.zoe_weights().items() if not k.startswith("core.")}
    t = lambda k: torch.from_numpy(z[k])
    with torch.no_grad():
        b0 = Z.seed_bin_regressor(w, t("x"))
        e0 = Z.projector(w, "seed_projector", t("x"))
        assert np.array_equal(b0.numpy(), z["seed_bins"]) and np.array_equal(e0.numpy(), z["seed_emb"])
        b1 = Z.attractor(w, "attractors.0", t("xb"), b0, e0)
        assert np.array_equal(b1.numpy(), z["bins0"])
        b3 = Z.attractor(w, "attractors.3", t("xb3"), b1, t("xb"))
        assert np.array_equal(b3.numpy(), z["bins3"])
        p = Z.conditional_log_binomial(w, t("last"), t("cond"))
        assert np.array_equal(p.numpy(), z["prob"])
        assert np.allclose(p.sum(1).numpy(), 1.0, atol=1e-5)
    # the configured attractor_alpha = 1000 would give different bins: the layer really runs with inv_attractor's default 300
    dx = t("xb")[:, :16].unsqueeze(2)[..., :5, :6] - b0.unsqueeze(1)
    assert not torch.equal(Z.inv_attractor(dx), dx.div(1 + 1000.0 * dx.pow(2)))


def test_pillow_resize_is_reproduced_exactly():
    from PIL import Image
    rng = np.random.default_rng(1)
    a = (rng.standard_normal((49, 65)) * 2 + 6).astype(np.float32)
    for oh, ow in [(135, 240), (24, 31), (49, 90), (100, 65), (49, 65)]:
        ref = np.asarray(Image.fromarray(a).resize((ow, oh)))
        assert np.array_equal(Z.pil_resize_f32(a, oh, ow), ref), (oh, ow)


def test_preprocess_geometry():
    fr = synth.frames(1, 90, 160, seed=2)[0]
    x = Z.preprocess(fr)
    assert x.shape == (1, 3, 392, 518) and x.dtype == np.float32
    # align_corners=True keeps the four corners
    mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
    for (yy, xx), (sy, sx) in {(0, 0): (0, 0), (391, 517): (89, 159), (0, 517): (0, 159)}.items():
        assert np.allclose(x[0, :, yy, xx], (fr[sy, sx].astype(np.float32) / 255 - mean) / std, atol=1e-6)
