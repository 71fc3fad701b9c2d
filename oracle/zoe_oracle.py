"""ORACLE (test infrastructure, never the product path) for `depth_anything --metric indoor|outdoor`: the ZoeDepth
metric head over the Depth-Anything core.

CPU restatement in numpy + torch.nn.functional fp32.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it, as the checker.  Paths below are relative to /root/reference/bands/.

Parity status
  * The head's layers - SeedBinRegressorUnnormed, Projector (patchfusion/zoedepth/models/layers/localbins_layers.py),
    AttractorLayerUnnormed + inv_attractor (layers/attractor.py), ConditionalLogBinomial / LogBinomial
    (layers/dist_layers.py) - are PINNED: oracle/make_golden.py loads those reference files (they import torch only)
    with seeded weights and asserts equality; tests/golden/zoe_layers.npz.
  * The DPT core underneath is the pinned depth oracle (oracle/depth_oracle.py).
  * PIL's bicubic resize of the mode-"F" prediction (depth_anything.py:117-119) is PINNED against Pillow itself
    (tests/test_zoe_oracle.py runs the real Image.resize).
  * UNPINNED: ZoeDepth.forward's own wiring (zoedepth_v1.py:139-215) and DepthAnythingCore (base_models/
    depth_anything.py:176-275: resize to 392x518 with align_corners=True, torchvision Normalize, forward hooks) cannot
    be imported (torchvision is absent); they are restated here line by line.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import depth_oracle as D

N_BINS = 64
# attractor.py:196-207 calls `dist(A - b)` WITHOUT alpha / gamma, so the function defaults (alpha 300, gamma 2) apply and the
# configured attractor_alpha = 1000 (config_zoedepth.json) is never used - pinned by oracle/make_golden.py zoe
ALPHA, GAMMA = 300.0, 2
MIN_TEMP, MAX_TEMP, P_EPS = 0.0212, 50.0, 1e-4
NET_H, NET_W = 392, 518


def _t(w, k):
    return torch.from_numpy(np.ascontiguousarray(w[k], dtype=np.float32))


def _c1(w, p, x):
    return F.conv2d(x, _t(w, p + ".weight"), _t(w, p + ".bias"))


def preprocess(frame_u8: np.ndarray) -> np.ndarray:
    """depth_anything.py:108-113 (PIL -> ToTensor: uint8 / 255 in float32) + DepthAnythingCore.prep
    (base_models/depth_anything.py:176-191): bilinear resize to 392 x 518 with align_corners=True (keep_aspect_ratio is
    False for the 'eval' config the band asks for), then Normalize(mean, std)."""
    x = torch.from_numpy(np.ascontiguousarray(frame_u8)).permute(2, 0, 1)[None].to(torch.float32).div(255)
    x = F.interpolate(x, (NET_H, NET_W), mode="bilinear", align_corners=True)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    return ((x - mean) / std).numpy()


def seed_bin_regressor(w, x):
    """SeedBinRegressorUnnormed.forward (localbins_layers.py:86-92): conv-ReLU-conv-Softplus."""
    return F.softplus(_c1(w, "seed_bin_regressor._net.2", F.relu(_c1(w, "seed_bin_regressor._net.0", x))))


def projector(w, p, x):
    """Projector.forward (localbins_layers.py:112-113)."""
    return _c1(w, p + "._net.2", F.relu(_c1(w, p + "._net.0", x)))


def inv_attractor(dx):
    """attractor.py:44-56: dc = dx / (1 + alpha dx^gamma)."""
    return dx.div(1 + ALPHA * dx.pow(GAMMA))


def attractor(w, p, x, b_prev, prev_b_embedding):
    """AttractorLayerUnnormed.forward (attractor.py:168-208), kind 'mean', type 'inv', not memory efficient."""
    prev = F.interpolate(prev_b_embedding, x.shape[-2:], mode="bilinear", align_corners=True)
    x = x + prev
    A = F.softplus(_c1(w, p + "._net.2", F.relu(_c1(w, p + "._net.0", x))))
    b = F.interpolate(b_prev, A.shape[-2:], mode="bilinear", align_corners=True)
    delta = torch.mean(inv_attractor(A.unsqueeze(2) - b.unsqueeze(1)), dim=1)
    return b + delta


def log_binom(n, k, eps=1e-7):
    """dist_layers.py:29-33, Stirling."""
    n = n + eps
    k = k + eps
    return n * torch.log(n) - k * torch.log(k) - (n - k) * torch.log(n - k + eps)


def conditional_log_binomial(w, x, cond):
    """ConditionalLogBinomial.forward (dist_layers.py:98-117) + LogBinomial.forward (:51-67), softmax over 64 bins."""
    pt = F.softplus(_c1(w, "conditional_log_binomial.mlp.2", F.gelu(_c1(w, "conditional_log_binomial.mlp.0", torch.cat((x, cond), 1)))))
    p, t = pt[:, :2], pt[:, 2:]
    p = p + P_EPS
    p = p[:, 0] / (p[:, 0] + p[:, 1])
    t = t + P_EPS
    t = t[:, 0] / (t[:, 0] + t[:, 1])
    t = t.unsqueeze(1)
    t = (MAX_TEMP - MIN_TEMP) * t + MIN_TEMP
    xx = p.unsqueeze(1)
    eps = 1e-4
    one_minus_x = torch.clamp(1 - xx, eps, 1)
    xx = torch.clamp(xx, eps, 1)
    k_idx = torch.arange(0, N_BINS).view(1, -1, 1, 1)
    K1 = torch.Tensor([N_BINS - 1]).view(1, -1, 1, 1)
    y = log_binom(K1, k_idx) + k_idx * torch.log(xx) + (N_BINS - 1 - k_idx) * torch.log(one_minus_x)
    return torch.softmax(y / t, dim=1)


def head(w, rel_depth, out_conv_act, l4_rn, blocks, return_stages=False):
    """ZoeDepth.forward after the core (zoedepth_v1.py:160-205).  blocks = [r4, r3, r2, r1]."""
    x = _c1(w, "conv2", l4_rn)
    b_prev = seed_bin_regressor(w, x)
    prev_emb = projector(w, "seed_projector", x)
    st = {"seed_bins": b_prev, "seed_emb": prev_emb}
    for i, xb in enumerate(blocks):
        emb = projector(w, f"projectors.{i}", xb)
        b = attractor(w, f"attractors.{i}", emb, b_prev, prev_emb)
        b_prev, prev_emb = b, emb
        st[f"bins{i}"] = b
    last = out_conv_act
    rel_cond = F.interpolate(rel_depth.unsqueeze(1), size=last.shape[2:], mode="bilinear", align_corners=True)
    last = torch.cat([last, rel_cond], dim=1)
    emb = F.interpolate(prev_emb, last.shape[-2:], mode="bilinear", align_corners=True)
    prob = conditional_log_binomial(w, last, emb)
    centers = F.interpolate(b_prev, prob.shape[-2:], mode="bilinear", align_corners=True)
    out = torch.sum(prob * centers, dim=1, keepdim=True)
    if return_stages:
        st.update(prob=prob, centers=centers)
        return out, st
    return out


def forward(w: Dict[str, np.ndarray], x: np.ndarray, return_stages: bool = False):
    """ZoeDepth.forward on the prepared input [B, 3, 392, 518] -> metric depth [B, 392, 518]."""
    core = {k[len("core.core."):]: v for k, v in w.items() if k.startswith("core.core.")}
    rel, st = D.model_forward(core, x, depth=24, heads=16, return_stages=True)
    tt = lambda k: torch.from_numpy(st[k])
    with torch.no_grad():
        res = head(w, torch.from_numpy(rel), tt("output_conv2_0"), tt("layer4_rn"), [tt("path4"), tt("path3"), tt("path2"), tt("path1")],
                   return_stages)
    if return_stages:
        out, hs = res
        st.update({k: v.numpy() for k, v in hs.items()})
        st["rel_depth"] = rel
        return out[:, 0].numpy(), st
    return res[:, 0].numpy()


# ---------------------------------------------------------------------------------------------------
# Pillow's Image.resize(size) of a mode-"F" image: default filter BICUBIC (a = -0.5), support scaled by the
# down-scaling factor, coefficients in double normalised to sum 1, horizontal pass then vertical pass with a float32
# intermediate (libImaging/Resample.c: precompute_coeffs, ImagingResampleHorizontal_32bpc / Vertical_32bpc).
# ---------------------------------------------------------------------------------------------------
def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_coeffs(in_size: int, out_size: int):
    """-> (xmin [out], count [out], coeffs [out, kmax] float64), as precompute_coeffs with box (0, in_size)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin_a = np.zeros(out_size, np.int32)
    cnt_a = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        for x in range(xmax):
            v = _bicubic((x + xmin - center + 0.5) * ss)
            kk[xx, x] = v
            ww += v
        for x in range(xmax):
            if ww != 0.0:
                kk[xx, x] /= ww
        xmin_a[xx], cnt_a[xx] = xmin, xmax
    return xmin_a, cnt_a, kk


def pil_resize_f32(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Image.fromarray(img).resize((out_w, out_h)) for float32 HxW (depth_anything.py:117-119)."""
    h, w = img.shape
    cur = img.astype(np.float32)
    if out_w != w:
        xmin, cnt, kk = pil_coeffs(w, out_w)
        tmp = np.empty((h, out_w), np.float32)
        for xx in range(out_w):
            acc = np.zeros(h, np.float64)
            for x in range(cnt[xx]):
                acc += cur[:, xmin[xx] + x].astype(np.float64) * kk[xx, x]
            tmp[:, xx] = acc.astype(np.float32)
        cur = tmp
    if out_h != h:
        ymin, cnt, kk = pil_coeffs(h, out_h)
        tmp = np.empty((out_h, cur.shape[1]), np.float32)
        for yy in range(out_h):
            acc = np.zeros(cur.shape[1], np.float64)
            for y in range(cnt[yy]):
                acc += cur[ymin[yy] + y].astype(np.float64) * kk[yy, y]
            tmp[yy] = acc.astype(np.float32)
        cur = tmp
    return cur


def infer(w, frame_u8: np.ndarray) -> np.ndarray:
    """bands/depth_anything.py:106-119: uint8 HxWx3 -> float32 HxW metric depth."""
    H, W = frame_u8.shape[:2]
    d = forward(w, preprocess(frame_u8))[0]
    return pil_resize_f32(d, H, W)
