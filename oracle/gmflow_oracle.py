"""ORACLE (test infrastructure, never the product path) for the flow_gmflow band.

CPU restatement in numpy + torch.nn.functional fp32 of /root/reference/bands/flow_gmflow.py and the GMFlow model it drives at the
band's defaults (feature_channels 128, num_scales 1, upsample_factor 8, 1 head, swin attention with attn_splits_list [2],
corr_radius_list [-1] = global matching, prop_radius_list [-1] = global propagation, 6 transformer blocks, ffn x 4,
padding_factor 16).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it, as the checker.  Pinned
against the imported reference modules by oracle/make_golden.py (vectors in tests/golden/gmflow_*.npz).

Parity status
  * GMFlow.forward (CNNEncoder, sine position embedding per window, FeatureTransformer with shifted windows, global correlation
    softmax, FeatureFlowAttention, convex upsampling), InputPadder(padding_factor=16): PINNED.
  * cv2.resize(frame, fx=fy=0.75, INTER_CUBIC) on uint8 (bands/flow_gmflow.py:151-152): PARITY UNPINNED, shared with
    oracle/raft_oracle.py (`cv_resize_cubic_u8`).

Paths cited are relative to /root/reference/.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .raft_oracle import cv_resize_cubic_u8


def _t(w, k):
    return torch.from_numpy(np.ascontiguousarray(w[k]))


def pad_amounts(h: int, w: int, factor: int = 16):
    """bands/common/flow.py:46-51 InputPadder(mode='sintel', padding_factor=16): [left, right, top, bottom]."""
    ph = (((h // factor) + 1) * factor - h) % factor
    pw = (((w // factor) + 1) * factor - w) % factor
    return [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]


# ---------------------------------------------------------------------------
# backbone (bands/gmflow/backbone.py)
# ---------------------------------------------------------------------------
def _inorm(x):
    return F.instance_norm(x, eps=1e-5)                      # nn.InstanceNorm2d defaults: no affine, no running stats


def residual_block(w, p: str, x, stride: int):
    """backbone.py:28-36: relu(norm1(conv1)), relu(norm2(conv2)), x -> downsample (1x1 conv with bias + norm3), relu(x + y)."""
    y = torch.relu(_inorm(F.conv2d(x, _t(w, p + "conv1.weight"), None, stride, 1)))
    y = torch.relu(_inorm(F.conv2d(y, _t(w, p + "conv2.weight"), None, 1, 1)))
    if (p + "downsample.0.weight") in w:
        x = _inorm(F.conv2d(x, _t(w, p + "downsample.0.weight"), _t(w, p + "downsample.0.bias"), stride, 0))
    return torch.relu(x + y)


def backbone(w, x):
    """CNNEncoder.forward with num_output_scales = 1 (backbone.py:102-117): [N, 3, H, W] -> [N, 128, H/8, W/8]."""
    x = torch.relu(_inorm(F.conv2d(x, _t(w, "backbone.conv1.weight"), None, 2, 3)))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = residual_block(w, f"backbone.layer{li}.0.", x, stride)
        x = residual_block(w, f"backbone.layer{li}.1.", x, 1)
    return F.conv2d(x, _t(w, "backbone.conv2.weight"), _t(w, "backbone.conv2.bias"))


# ---------------------------------------------------------------------------
# position embedding (bands/gmflow/position.py:26-46, utils.py:61-86)
# ---------------------------------------------------------------------------
def position_sine(h: int, w: int, channels: int = 128):
    """PositionEmbeddingSine(num_pos_feats = channels / 2, temperature 10000, normalize, scale 2 pi) of an h x w map: [channels, h, w]."""
    npf = channels // 2
    y_embed = torch.arange(1, h + 1, dtype=torch.float32)[:, None].repeat(1, w)
    x_embed = torch.arange(1, w + 1, dtype=torch.float32)[None, :].repeat(h, 1)
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[-1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, -1:] + eps) * scale
    dim_t = torch.arange(npf, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / npf)
    pos_x = x_embed[:, :, None] / dim_t
    pos_y = y_embed[:, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, 0::2].sin(), pos_x[:, :, 1::2].cos()), dim=3).flatten(2)
    pos_y = torch.stack((pos_y[:, :, 0::2].sin(), pos_y[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((pos_y, pos_x), dim=2).permute(2, 0, 1)


def add_position(f0, f1, splits: int):
    """feature_add_position (utils.py:61-86): with attn_splits > 1 the embedding of ONE window is tiled over the splits x splits windows."""
    b, c, h, w = f0.shape
    pos = position_sine(h // splits, w // splits, c).repeat(1, splits, splits) if splits > 1 else position_sine(h, w, c)
    return f0 + pos[None], f1 + pos[None]


# ---------------------------------------------------------------------------
# transformer (bands/gmflow/transformer.py)
# ---------------------------------------------------------------------------
def split_windows(x, k: int):
    """utils.split_feature(channel_last=True): [B, H, W, C] -> [B k k, H/k, W/k, C]."""
    b, h, w, c = x.shape
    return x.view(b, k, h // k, k, w // k, c).permute(0, 1, 3, 2, 4, 5).reshape(b * k * k, h // k, w // k, c)


def merge_windows(x, k: int):
    """utils.merge_splits(channel_last=True)."""
    bk, h, w, c = x.shape
    b = bk // (k * k)
    return x.view(b, k, k, h, w, c).permute(0, 1, 3, 2, 4, 5).reshape(b, k * h, k * w, c)


def shift_mask(h: int, w: int, wh: int, ww: int):
    """generate_shift_window_attn_mask (transformer.py:18-44): [K*K, wh*ww, wh*ww] of 0 / -100."""
    sh, sw = wh // 2, ww // 2
    img = torch.zeros((1, h, w, 1))
    cnt = 0
    for hs in (slice(0, -wh), slice(-wh, -sh), slice(-sh, None)):
        for ws in (slice(0, -ww), slice(-ww, -sw), slice(-sw, None)):
            img[:, hs, ws, :] = cnt
            cnt += 1
    mw = split_windows(img, w // ww).view(-1, wh * ww)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def window_attention(q, k, v, splits: int, shifted: bool, h: int, w: int, mask):
    """single_head_split_window_attention (transformer.py:47-101): q, k, v [B, L, C]."""
    b, _, c = q.shape
    wh, ww = h // splits, w // splits
    q, k, v = (t.view(b, h, w, c) for t in (q, k, v))
    if shifted:
        q, k, v = (torch.roll(t, shifts=(-(wh // 2), -(ww // 2)), dims=(1, 2)) for t in (q, k, v))
    q, k, v = (split_windows(t, splits).reshape(b * splits * splits, wh * ww, c) for t in (q, k, v))
    scores = torch.matmul(q, k.permute(0, 2, 1)) / (c ** 0.5)
    if shifted:
        scores = scores + mask.repeat(b, 1, 1)
    out = torch.matmul(torch.softmax(scores, dim=-1), v)
    out = merge_windows(out.view(b * splits * splits, wh, ww, c), splits)
    if shifted:
        out = torch.roll(out, shifts=(wh // 2, ww // 2), dims=(1, 2))
    return out.reshape(b, h * w, c)


def transformer_layer(w, p: str, source, target, h, wd, splits, shifted, mask, ffn: bool):
    """TransformerLayer.forward (transformer.py:141-181)."""
    q = F.linear(source, _t(w, p + "q_proj.weight"))
    k = F.linear(target, _t(w, p + "k_proj.weight"))
    v = F.linear(target, _t(w, p + "v_proj.weight"))
    if splits > 1:
        msg = window_attention(q, k, v, splits, shifted, h, wd, mask)
    else:
        msg = torch.matmul(torch.softmax(torch.matmul(q, k.permute(0, 2, 1)) / (q.shape[2] ** 0.5), dim=2), v)
    c = source.shape[-1]
    msg = F.layer_norm(F.linear(msg, _t(w, p + "merge.weight")), (c,), _t(w, p + "norm1.weight"), _t(w, p + "norm1.bias"))
    if ffn:
        msg = F.linear(F.gelu(F.linear(torch.cat([source, msg], dim=-1), _t(w, p + "mlp.0.weight"))), _t(w, p + "mlp.2.weight"))
        msg = F.layer_norm(msg, (c,), _t(w, p + "norm2.weight"), _t(w, p + "norm2.bias"))
    return source + msg


def feature_transformer(w, f0, f1, splits: int, layers: int = 6, stages=None):
    """FeatureTransformer.forward (transformer.py:248-290): both directions as one batch [f0; f1] attending to [f1; f0]."""
    b, c, h, wd = f0.shape
    a = f0.flatten(-2).permute(0, 2, 1)
    bb = f1.flatten(-2).permute(0, 2, 1)
    mask = shift_mask(h, wd, h // splits, wd // splits) if splits > 1 else None
    c0, c1 = torch.cat((a, bb), 0), torch.cat((bb, a), 0)
    for i in range(layers):
        shifted = splits > 1 and i % 2 == 1
        p = f"transformer.layers.{i}."
        c0 = transformer_layer(w, p + "self_attn.", c0, c0, h, wd, splits, shifted, mask, False)
        c0 = transformer_layer(w, p + "cross_attn_ffn.", c0, c1, h, wd, splits, shifted, mask, True)
        c1 = torch.cat(c0.chunk(2, 0)[::-1], 0)
        if stages is not None and i in (0, layers - 1):
            stages[f"block{i}"] = c0.clone()
    a, bb = c0.chunk(2, 0)
    return a.view(b, h, wd, c).permute(0, 3, 1, 2).contiguous(), bb.view(b, h, wd, c).permute(0, 3, 1, 2).contiguous()


# ---------------------------------------------------------------------------
# matching, propagation, upsampling (bands/gmflow/matching.py:7-42, transformer.py:300-337, gmflow.py:66-93)
# ---------------------------------------------------------------------------
def coords_grid(b, h, w):
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack([x, y], 0).float()[None].repeat(b, 1, 1, 1)


def global_correlation_softmax(f0, f1, bidir: bool):
    b, c, h, w = f0.shape
    corr = torch.matmul(f0.view(b, c, -1).permute(0, 2, 1), f1.view(b, c, -1)) / (c ** 0.5)         # [B, HW, HW]
    grid0 = coords_grid(b, h, w)
    grid = grid0.view(b, 2, -1).permute(0, 2, 1)
    if bidir:
        corr = torch.cat((corr, corr.permute(0, 2, 1)), 0)
        grid0, grid, b = grid0.repeat(2, 1, 1, 1), grid.repeat(2, 1, 1), b * 2
    prob = F.softmax(corr, dim=-1)
    return torch.matmul(prob, grid).view(b, h, w, 2).permute(0, 3, 1, 2) - grid0


def flow_attention(w, f0, flow):
    """FeatureFlowAttention.forward, global (transformer.py:316-337).  The key is k_proj of the PROJECTED query, as in the reference."""
    b, c, h, wd = f0.shape
    q = F.linear(f0.view(b, c, h * wd).permute(0, 2, 1), _t(w, "feature_flow_attn.q_proj.weight"), _t(w, "feature_flow_attn.q_proj.bias"))
    k = F.linear(q, _t(w, "feature_flow_attn.k_proj.weight"), _t(w, "feature_flow_attn.k_proj.bias"))
    v = flow.view(b, 2, h * wd).permute(0, 2, 1)
    prob = torch.softmax(torch.matmul(q, k.permute(0, 2, 1)) / (c ** 0.5), dim=-1)
    return torch.matmul(prob, v).view(b, h, wd, 2).permute(0, 3, 1, 2)


def upsample_flow(w, flow, feature, factor: int = 8):
    """GMFlow.upsample_flow, convex (gmflow.py:74-92)."""
    x = torch.relu(F.conv2d(torch.cat((flow, feature), 1), _t(w, "upsampler.0.weight"), _t(w, "upsampler.0.bias"), 1, 1))
    mask = F.conv2d(x, _t(w, "upsampler.2.weight"), _t(w, "upsampler.2.bias"))
    b, _, h, wd = flow.shape
    mask = torch.softmax(mask.view(b, 1, 9, factor, factor, h, wd), dim=2)
    up = F.unfold(factor * flow, [3, 3], padding=1).view(b, 2, 9, 1, 1, h, wd)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(b, 2, factor * h, factor * wd)


def gmflow_forward(w: Dict[str, np.ndarray], img0: np.ndarray, img1: np.ndarray, bidir: bool = False, splits: int = 2,
                   return_stages: bool = False):
    """GMFlow.forward at the band's defaults (gmflow.py:95-170): float images [B, 3, H, W] in 0..255 (H, W multiples of 16).
    Returns flow_up [B or 2B, 2, H, W] (+ stages)."""
    st = {}
    with torch.no_grad():
        mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        i0 = (torch.from_numpy(np.ascontiguousarray(img0)).float() / 255.0 - mean) / std            # utils.py:53-58
        i1 = (torch.from_numpy(np.ascontiguousarray(img1)).float() / 255.0 - mean) / std
        feat = backbone(w, torch.cat((i0, i1), 0))
        f0, f1 = feat.chunk(2, 0)
        st["feat0"] = f0.clone()
        f0, f1 = add_position(f0, f1, splits)
        f0, f1 = feature_transformer(w, f0, f1, splits, stages=st)
        st["tfeat0"] = f0.clone()
        flow = global_correlation_softmax(f0, f1, bidir)
        st["flow_match"] = flow.clone()
        if bidir:
            f0 = torch.cat((f0, f1), 0)
        flow = flow_attention(w, f0, flow)
        st["flow_prop"] = flow.clone()
        up = upsample_flow(w, flow, f0)
    if return_stages:
        return up.numpy(), {k: v.numpy() for k, v in st.items()}
    return up.numpy()


def infer_pair(w, prev_u8: np.ndarray, curr_u8: np.ndarray, scale: float = 0.75, backward: bool = True, inference_size=None):
    """bands/flow_gmflow.py:149-157 + infer (:66-118): two uint8 frames -> (fwd, bwd or None) float32 [H', W', 2] at the scaled
    resolution (InputPadder(padding_factor=16), pred_bidir_flow when --backwards / masks are asked for).  inference_size (H, W)
    (:76-80, :92-97): no padding - bilinear (align_corners) to that size in, bilinear back out with u * W' / W and v * H' / H."""
    a = cv_resize_cubic_u8(prev_u8, scale) if scale != 1.0 else prev_u8
    c = cv_resize_cubic_u8(curr_u8, scale) if scale != 1.0 else curr_u8
    ta = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).float()[None]
    tc = torch.from_numpy(np.ascontiguousarray(c)).permute(2, 0, 1).float()[None]
    if inference_size is not None:
        ori = ta.shape[-2:]
        ia = F.interpolate(ta, size=tuple(inference_size), mode="bilinear", align_corners=True)
        ic = F.interpolate(tc, size=tuple(inference_size), mode="bilinear", align_corners=True)
        up = torch.from_numpy(gmflow_forward(w, ia.numpy(), ic.numpy(), bidir=backward))
        up = F.interpolate(up, size=tuple(ori), mode="bilinear", align_corners=True)
        up[:, 0] = up[:, 0] * ori[-1] / inference_size[-1]
        up[:, 1] = up[:, 1] * ori[-2] / inference_size[-2]
        up = up.numpy()
        return np.ascontiguousarray(up[0].transpose(1, 2, 0)), (np.ascontiguousarray(up[1].transpose(1, 2, 0)) if backward else None)
    pad = pad_amounts(ta.shape[2], ta.shape[3])
    pa, pc = F.pad(ta, pad, mode="replicate"), F.pad(tc, pad, mode="replicate")
    up = gmflow_forward(w, pa.numpy(), pc.numpy(), bidir=backward)
    ht, wd = up.shape[-2:]
    up = up[..., pad[2]:ht - pad[3], pad[0]:wd - pad[1]]
    fwd = np.ascontiguousarray(up[0].transpose(1, 2, 0))
    bwd = np.ascontiguousarray(up[1].transpose(1, 2, 0)) if backward else None
    return fwd, bwd
