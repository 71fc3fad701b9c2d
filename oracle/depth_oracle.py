"""ORACLE (test infrastructure, never the product path) for the depth_anything band.

A CPU restatement, in numpy + torch.nn.functional fp32, of what
/root/reference/bands/depth_anything.py does per frame.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module,
and only as the checker.  Pinned against the real reference modules by
oracle/make_golden.py (run in the build container, where /root/reference exists);
the resulting vectors live in tests/golden/.

Parity status
  * ViT + DPT head + band resize + heat encode: PINNED (oracle == imported reference
    modules on seeded weights, see make_golden.py / tests/test_oracle_golden.py).
  * cv2.resize(INTER_CUBIC) in the pre-process: PARITY UNPINNED - third-party
    opencv-python 4.8.1.78 (environment.yml) is absent here; `cv_resize_cubic`
    restates OpenCV's published bicubic (a = -0.75, half-pixel centres, replicated
    border, float32 coefficient table, no antialias).  Call site:
    bands/d_anything/util/transform.py:174-178.

Every function cites the reference lines it follows (paths relative to
/root/reference/).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

MEAN = np.array([0.485, 0.456, 0.406])      # bands/depth_anything.py:72
STD = np.array([0.229, 0.224, 0.225])


# ---------------------------------------------------------------------------
# pre-process
# ---------------------------------------------------------------------------
def net_size(width: int, height: int, target: int = 518, multiple: int = 14) -> Tuple[int, int]:
    """(net_w, net_h): keep-aspect 'lower_bound' resize, each side rounded to a multiple of 14.

    bands/d_anything/util/transform.py:100-109 (constrain_to_multiple_of) and :111-166
    (get_size) with the arguments of bands/depth_anything.py:63-71.
    """
    scale_h = target / height
    scale_w = target / width
    if scale_w > scale_h:
        scale_h = scale_w
    else:
        scale_w = scale_h

    def constrain(x: float, min_val: int) -> int:
        y = int(np.round(x / multiple) * multiple)
        if y < min_val:
            y = int(np.ceil(x / multiple) * multiple)
        return y

    return constrain(scale_w * width, target), constrain(scale_h * height, target)


def _cubic_coeffs(x: np.ndarray) -> np.ndarray:
    """OpenCV interpolateCubic, float32, A = -0.75 (imgproc/resize.cpp)."""
    A = np.float32(-0.75)
    x = x.astype(np.float32)
    one = np.float32(1.0)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
    c2 = ((A + np.float32(2)) * (one - x) - (A + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.float32)


def cubic_taps(src: int, dst: int) -> Tuple[np.ndarray, np.ndarray]:
    """Per destination index: 4 clamped source indices and 4 float32 weights."""
    scale = 1.0 / (dst / src)                       # OpenCV: scale_x = 1. / inv_scale_x
    d = np.arange(dst, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = fx - sx.astype(np.float32)
    idx = np.clip(sx[:, None] + np.arange(-1, 3)[None, :], 0, src - 1)
    return idx, _cubic_coeffs(fx)


def cv_resize_cubic(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_CUBIC) on float64 HxWxC."""
    h, w = img.shape[:2]
    xi, xw = cubic_taps(w, out_w)
    yi, yw = cubic_taps(h, out_h)
    img = img.astype(np.float64)
    # horizontal pass (double accumulation, float coefficients), then vertical
    tmp = np.zeros((h, out_w, img.shape[2]), np.float64)
    for t in range(4):
        tmp += img[:, xi[:, t], :] * xw[:, t].astype(np.float64)[None, :, None]
    out = np.zeros((out_h, out_w, img.shape[2]), np.float64)
    for t in range(4):
        out += tmp[yi[:, t], :, :] * yw[:, t].astype(np.float64)[:, None, None]
    return out


def preprocess(img_u8: np.ndarray) -> np.ndarray:
    """uint8 RGB HxWx3 -> float32 3 x net_h x net_w.

    bands/depth_anything.py:122-126: img/255.0 (float64) -> Resize -> NormalizeImage ->
    PrepareForNet (transform.py:168-178, 219-222, 232-234).
    """
    h, w = img_u8.shape[:2]
    nw, nh = net_size(w, h)
    x = img_u8 / 255.0
    x = cv_resize_cubic(x, nw, nh)
    x = (x - MEAN) / STD
    return np.ascontiguousarray(np.transpose(x, (2, 0, 1))).astype(np.float32)


# ---------------------------------------------------------------------------
# DINOv2 ViT
# ---------------------------------------------------------------------------
def _t(w: Dict[str, np.ndarray], k: str) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(w[k]))


def interp_pos_embed(pos_embed: np.ndarray, gh: int, gw: int, offset: float = 0.1) -> np.ndarray:
    """[1, 1+g*g, D] -> [1+gh*gw, D] float32.

    vision_transformer.py:179-210.  The reference names the tensor dims (w, h) =
    (H, W) of the image (prepare_tokens_with_masks :213 unpacks `B, nc, w, h`), builds
    the 37x37 grid as [rows, cols], and calls bicubic interpolate with
    scale_factor = ((gh+0.1)/37, (gw+0.1)/37); torch then maps
    src = (dst + 0.5) / scale_factor - 0.5, a = -0.75, clamped taps.
    """
    pe = torch.from_numpy(pos_embed).float()
    n = pe.shape[1] - 1
    g = int(math.sqrt(n))
    dim = pe.shape[-1]
    if gh * gw == n and gh == gw:
        return pe[0].numpy()
    patch = pe[:, 1:].reshape(1, g, g, dim).permute(0, 3, 1, 2)
    sy, sx = float(gh + offset) / g, float(gw + offset) / g
    out = F.interpolate(patch, scale_factor=(sy, sx), mode="bicubic")
    assert out.shape[-2] == gh and out.shape[-1] == gw
    out = out.permute(0, 2, 3, 1).reshape(-1, dim)
    return torch.cat([pe[0, :1], out], 0).numpy()


def vit_features(w: Dict[str, np.ndarray], x: torch.Tensor, depth: int, heads: int,
                 n_taps: int = 4, return_stages: bool = False):
    """x [B,3,H,W] float32 -> list of n_taps tensors [B, gh*gw, D] (normed, cls dropped).

    patch_embed (dinov2/layers/patch_embed.py:66-82), tokens + pos-embed
    (vision_transformer.py:212-231), blocks (dinov2/layers/block.py:82-107 eval branch,
    attention.py:49-62, mlp.py:35-41, layer_scale.py:27-28), taps
    (vision_transformer.py:271-281, 297-321; called with n=4 at d_anything/dpt.py:158).
    """
    P = "pretrained."
    B, _, H, W = x.shape
    gh, gw = H // 14, W // 14
    stages = {}
    t = F.conv2d(x, _t(w, P + "patch_embed.proj.weight"), _t(w, P + "patch_embed.proj.bias"), stride=14)
    t = t.flatten(2).transpose(1, 2)                                   # B, gh*gw, D
    D = t.shape[-1]
    stages["patch_embed"] = t
    cls = _t(w, P + "cls_token").expand(B, -1, -1)
    t = torch.cat([cls, t], 1)
    t = t + torch.from_numpy(interp_pos_embed(w[P + "pos_embed"], gh, gw))[None]
    stages["tokens"] = t
    hd = D // heads
    taps = []
    for i in range(depth):
        p = f"{P}blocks.{i}."
        y = F.layer_norm(t, (D,), _t(w, p + "norm1.weight"), _t(w, p + "norm1.bias"), eps=1e-6)
        qkv = F.linear(y, _t(w, p + "attn.qkv.weight"), _t(w, p + "attn.qkv.bias"))
        qkv = qkv.reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
        a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        y = (a @ v).transpose(1, 2).reshape(B, -1, D)
        y = F.linear(y, _t(w, p + "attn.proj.weight"), _t(w, p + "attn.proj.bias"))
        t = t + y * _t(w, p + "ls1.gamma")
        y = F.layer_norm(t, (D,), _t(w, p + "norm2.weight"), _t(w, p + "norm2.bias"), eps=1e-6)
        y = F.linear(y, _t(w, p + "mlp.fc1.weight"), _t(w, p + "mlp.fc1.bias"))
        y = F.gelu(y)                                                   # nn.GELU() = exact erf
        y = F.linear(y, _t(w, p + "mlp.fc2.weight"), _t(w, p + "mlp.fc2.bias"))
        t = t + y * _t(w, p + "ls2.gamma")
        stages[f"block{i}"] = t
        if i >= depth - n_taps:
            taps.append(t)
    nw, nb = _t(w, P + "norm.weight"), _t(w, P + "norm.bias")
    feats = [F.layer_norm(o, (D,), nw, nb, eps=1e-6)[:, 1:] for o in taps]
    if return_stages:
        return feats, stages
    return feats


# ---------------------------------------------------------------------------
# DPT head
# ---------------------------------------------------------------------------
def _rcu(w, p: str, x: torch.Tensor) -> torch.Tensor:
    """ResidualConvUnit, bn=False, activation ReLU (not in-place). d_anything/blocks.py:69-92."""
    o = F.relu(x)
    o = F.conv2d(o, _t(w, p + "conv1.weight"), _t(w, p + "conv1.bias"), padding=1)
    o = F.relu(o)
    o = F.conv2d(o, _t(w, p + "conv2.weight"), _t(w, p + "conv2.bias"), padding=1)
    return o + x


def _fusion(w, p: str, x0: torch.Tensor, x1, size) -> torch.Tensor:
    """FeatureFusionBlock.forward. d_anything/blocks.py:126-153 (align_corners=True, dpt.py:10-19)."""
    o = x0
    if x1 is not None:
        o = o + _rcu(w, p + "resConfUnit1.", x1)
    o = _rcu(w, p + "resConfUnit2.", o)
    if size is None:
        o = F.interpolate(o, scale_factor=2, mode="bilinear", align_corners=True)
    else:
        o = F.interpolate(o, size=size, mode="bilinear", align_corners=True)
    return F.conv2d(o, _t(w, p + "out_conv.weight"), _t(w, p + "out_conv.bias"))


def dpt_head(w: Dict[str, np.ndarray], feats: List[torch.Tensor], gh: int, gw: int,
             return_stages: bool = False):
    """DPTHead.forward, use_clstoken=False. d_anything/dpt.py:103-136."""
    h = "depth_head."
    st = {}
    outs = []
    for i, x in enumerate(feats):
        B, _, D = x.shape
        x = x.permute(0, 2, 1).reshape(B, D, gh, gw)
        x = F.conv2d(x, _t(w, h + f"projects.{i}.weight"), _t(w, h + f"projects.{i}.bias"))
        if i == 0:
            x = F.conv_transpose2d(x, _t(w, h + "resize_layers.0.weight"), _t(w, h + "resize_layers.0.bias"), stride=4)
        elif i == 1:
            x = F.conv_transpose2d(x, _t(w, h + "resize_layers.1.weight"), _t(w, h + "resize_layers.1.bias"), stride=2)
        elif i == 3:
            x = F.conv2d(x, _t(w, h + "resize_layers.3.weight"), _t(w, h + "resize_layers.3.bias"), stride=2, padding=1)
        st[f"layer{i + 1}"] = x
        outs.append(x)
    rn = [F.conv2d(o, _t(w, h + f"scratch.layer{i + 1}_rn.weight"), None, padding=1) for i, o in enumerate(outs)]
    for i in range(4):
        st[f"layer{i + 1}_rn"] = rn[i]
    s = h + "scratch."
    p4 = _fusion(w, s + "refinenet4.", rn[3], None, rn[2].shape[2:])
    p3 = _fusion(w, s + "refinenet3.", p4, rn[2], rn[1].shape[2:])
    p2 = _fusion(w, s + "refinenet2.", p3, rn[1], rn[0].shape[2:])
    p1 = _fusion(w, s + "refinenet1.", p2, rn[0], None)
    st.update(path4=p4, path3=p3, path2=p2, path1=p1)
    o = F.conv2d(p1, _t(w, s + "output_conv1.weight"), _t(w, s + "output_conv1.bias"), padding=1)
    st["output_conv1"] = o
    o = F.interpolate(o, (gh * 14, gw * 14), mode="bilinear", align_corners=True)
    o = F.relu(F.conv2d(o, _t(w, s + "output_conv2.0.weight"), _t(w, s + "output_conv2.0.bias"), padding=1))
    st["output_conv2_0"] = o
    o = F.conv2d(o, _t(w, s + "output_conv2.2.weight"), _t(w, s + "output_conv2.2.bias"))
    st["pre_relu"] = o
    o = F.relu(o)
    if return_stages:
        return o, st
    return o


def model_forward(w: Dict[str, np.ndarray], x: np.ndarray, depth: int, heads: int,
                 return_stages: bool = False):
    """DPT_DINOv2.forward: [B,3,h,w] -> [B,h,w]. d_anything/dpt.py:155-166."""
    with torch.no_grad():
        xt = torch.from_numpy(x)
        hgt, wid = xt.shape[-2:]
        gh, gw = hgt // 14, wid // 14
        if return_stages:
            feats, st1 = vit_features(w, xt, depth, heads, return_stages=True)
            d, st2 = dpt_head(w, feats, gh, gw, return_stages=True)
            st1.update(st2)
            for i, f in enumerate(feats):
                st1[f"feat{i}"] = f
        else:
            feats = vit_features(w, xt, depth, heads)
            d = dpt_head(w, feats, gh, gw)
        d = F.interpolate(d, size=(hgt, wid), mode="bilinear", align_corners=True)
        d = F.relu(d).squeeze(1)
    if return_stages:
        return d.numpy(), {k: v.numpy() for k, v in st1.items()}
    return d.numpy()


def infer(w: Dict[str, np.ndarray], img_u8: np.ndarray, depth: int, heads: int) -> np.ndarray:
    """depth_anything.infer(img) relative branch: uint8 RGB HxWx3 -> float32 HxW.

    bands/depth_anything.py:121-133.
    """
    h, wd = img_u8.shape[:2]
    x = preprocess(img_u8)[None]
    d = model_forward(w, x, depth, heads)
    with torch.no_grad():
        d = F.interpolate(torch.from_numpy(d)[None], (h, wd), mode="bilinear", align_corners=False)[0, 0]
    return d.numpy()


# ---------------------------------------------------------------------------
# post-process (video path)
# ---------------------------------------------------------------------------
def hue_to_rgb(hue: np.ndarray) -> np.ndarray:
    """bands/common/encode.py:13-28 (float64)."""
    hue = np.asarray(hue, dtype=np.float64)
    rgb = np.stack([hue * 6.0, hue * 6.0 + 4.0, hue * 6.0 + 2.0], axis=-1)
    rgb = np.abs(np.mod(rgb, 6.0) - 3.0) - 1.0
    return np.clip(rgb, 0.0, 1.0)


def heat_to_rgb(heat: np.ndarray) -> np.ndarray:
    """bands/common/encode.py:31-33."""
    return hue_to_rgb((1.0 - np.asarray(heat, np.float64)) * 0.65)


def encode_depth_video(pred: np.ndarray, flip: bool = True):
    """float32 HxW -> (uint8 HxWx3, min, max). bands/depth_anything.py:215-221.

    min/max/normalise/flip run in float32 (numpy keeps the prediction dtype), the
    colour ramp in float64, uint8 conversion truncates.
    """
    pred = np.asarray(pred, np.float32)
    dmin = pred.min()
    dmax = pred.max()
    with np.errstate(invalid="ignore", divide="ignore"):
        d = (pred - dmin) / (dmax - dmin)
        if flip:
            d = np.float32(1.0) - d
        rgb = (heat_to_rgb(d.astype(np.float64)) * 255).astype(np.uint8)
    return rgb, float(dmin), float(dmax)


def rgb_to_heat(rgb_u8: np.ndarray) -> np.ndarray:
    """Decode contract of the viewer (bands/common/encode.py:36-64 rgb_to_hsv / rgb_to_heat, used by
    view.py:186-210): uint8 heat image -> heat in 0..1."""
    rgb = rgb_u8.astype("float")
    maxv, maxc = rgb.max(axis=2), rgb.argmax(axis=2)
    minv, minc = rgb.min(axis=2), rgb.argmin(axis=2)
    eps = np.spacing(1)
    hue = np.zeros(maxv.shape)
    h0 = ((rgb[..., 1] - rgb[..., 2]) * 60.0 / (maxv - minv + eps)) % 360.0
    h1 = (rgb[..., 2] - rgb[..., 0]) * 60.0 / (maxv - minv + eps) + 120.0
    h2 = (rgb[..., 0] - rgb[..., 1]) * 60.0 / (maxv - minv + eps) + 240.0
    hue[maxc == 0] = h0[maxc == 0]
    hue[maxc == 1] = h1[maxc == 1]
    hue[maxc == 2] = h2[maxc == 2]
    hue[maxc == minc] = 0.0
    return np.clip(1.0 - hue / 360.0 * 1.538461538, 0.0, 1.0)
