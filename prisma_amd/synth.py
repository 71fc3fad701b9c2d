"""Seeded synthetic weights and frames for the bands engine.

There is no network on the build or GPU boxes, so neither the Depth-Anything
nor the RAFT checkpoints can be fetched.  This module generates weights with
the *same key names and shapes* as the reference checkpoints
(Depth-Anything: `pretrained.*` / `depth_head.*`, see
/root/reference/bands/d_anything/dpt.py:139-166; RAFT: `fnet.* cnet.*
update_block.*`, see /root/reference/bands/raft/raft.py:24-58) from a numpy
PCG64 stream keyed by (seed, crc32(param name)).  The stream only depends on
the name, so the oracle script (which loads the tensors into the reference
torch modules), the tests and bench.py on the GPU box all see bit-identical
weights without shipping a checkpoint.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np


# ----------------------------------------------------------------------------
# Depth-Anything (DINOv2 ViT + DPT head) configurations.
# Encoder table: reference hubconf vit_small/base/large
# (/root/reference/bands/d_anything/torchhub/facebookresearch_dinov2_main/vision_transformer.py:340-378)
# DPT table: the HF configs used by DepthAnything.from_pretrained
# (LiheYoung/depth_anything_vit{s,b,l}14: features / out_channels).
# ----------------------------------------------------------------------------
@dataclass(frozen=True)
class DepthCfg:
    name: str = "vitl"
    embed_dim: int = 1024
    depth: int = 24
    heads: int = 16
    mlp_ratio: int = 4
    patch: int = 14
    pos_grid: int = 37                      # 518 / 14, pos_embed has 1 + 37*37 rows
    features: int = 256
    out_channels: Tuple[int, int, int, int] = (256, 512, 1024, 1024)
    n_taps: int = 4                         # get_intermediate_layers(x, 4)

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.heads


DEPTH_CFGS: Dict[str, DepthCfg] = {
    "vits": DepthCfg("vits", 384, 12, 6, 4, 14, 37, 64, (48, 96, 192, 384)),
    "vitb": DepthCfg("vitb", 768, 12, 12, 4, 14, 37, 128, (96, 192, 384, 768)),
    "vitl": DepthCfg("vitl", 1024, 24, 16, 4, 14, 37, 256, (256, 512, 1024, 1024)),
    # ViT-L widths with only 4 blocks: same kernels/shapes as vitl, cheap on CPU.
    "vitl_d4": DepthCfg("vitl_d4", 1024, 4, 16, 4, 14, 37, 256, (256, 512, 1024, 1024)),
}


def depth_param_shapes(cfg: DepthCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    D = cfg.embed_dim
    H = D * cfg.mlp_ratio
    F = cfg.features
    oc = cfg.out_channels
    out: List[Tuple[str, Tuple[int, ...]]] = [
        ("pretrained.cls_token", (1, 1, D)),
        ("pretrained.pos_embed", (1, 1 + cfg.pos_grid * cfg.pos_grid, D)),
        ("pretrained.mask_token", (1, D)),
        ("pretrained.patch_embed.proj.weight", (D, 3, cfg.patch, cfg.patch)),
        ("pretrained.patch_embed.proj.bias", (D,)),
    ]
    for i in range(cfg.depth):
        p = f"pretrained.blocks.{i}."
        out += [
            (p + "norm1.weight", (D,)), (p + "norm1.bias", (D,)),
            (p + "attn.qkv.weight", (3 * D, D)), (p + "attn.qkv.bias", (3 * D,)),
            (p + "attn.proj.weight", (D, D)), (p + "attn.proj.bias", (D,)),
            (p + "ls1.gamma", (D,)),
            (p + "norm2.weight", (D,)), (p + "norm2.bias", (D,)),
            (p + "mlp.fc1.weight", (H, D)), (p + "mlp.fc1.bias", (H,)),
            (p + "mlp.fc2.weight", (D, H)), (p + "mlp.fc2.bias", (D,)),
            (p + "ls2.gamma", (D,)),
        ]
    out += [("pretrained.norm.weight", (D,)), ("pretrained.norm.bias", (D,))]
    h = "depth_head."
    for i in range(4):
        out += [(h + f"projects.{i}.weight", (oc[i], D, 1, 1)), (h + f"projects.{i}.bias", (oc[i],))]
    out += [
        (h + "resize_layers.0.weight", (oc[0], oc[0], 4, 4)), (h + "resize_layers.0.bias", (oc[0],)),
        (h + "resize_layers.1.weight", (oc[1], oc[1], 2, 2)), (h + "resize_layers.1.bias", (oc[1],)),
        (h + "resize_layers.3.weight", (oc[3], oc[3], 3, 3)), (h + "resize_layers.3.bias", (oc[3],)),
    ]
    for i in range(4):
        out.append((h + f"scratch.layer{i + 1}_rn.weight", (F, oc[i], 3, 3)))
    for r in range(1, 5):
        p = h + f"scratch.refinenet{r}."
        out += [(p + "out_conv.weight", (F, F, 1, 1)), (p + "out_conv.bias", (F,))]
        for u in (1, 2):
            for c in (1, 2):
                out += [(p + f"resConfUnit{u}.conv{c}.weight", (F, F, 3, 3)),
                        (p + f"resConfUnit{u}.conv{c}.bias", (F,))]
    out += [
        (h + "scratch.output_conv1.weight", (F // 2, F, 3, 3)), (h + "scratch.output_conv1.bias", (F // 2,)),
        (h + "scratch.output_conv2.0.weight", (32, F // 2, 3, 3)), (h + "scratch.output_conv2.0.bias", (32,)),
        (h + "scratch.output_conv2.2.weight", (1, 32, 1, 1)), (h + "scratch.output_conv2.2.bias", (1,)),
    ]
    return out


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed & 0xFFFFFFFF, zlib.crc32(name.encode())])


def _fan_in(name: str, shape: Tuple[int, ...]) -> int:
    if len(shape) == 4:
        if "resize_layers.0" in name or "resize_layers.1" in name:   # ConvTranspose2d: (in, out, kh, kw)
            return shape[0]
        return shape[1] * shape[2] * shape[3]
    if len(shape) == 2:
        return shape[1]
    return 1


def _depth_tensor(seed: int, name: str, shape: Tuple[int, ...]) -> np.ndarray:
    g = _rng(seed, name)
    n = lambda s: g.standard_normal(shape, dtype=np.float32) * np.float32(s)
    if name.endswith("cls_token") or name.endswith("pos_embed"):
        return n(0.2)
    if name.endswith("mask_token"):
        return np.zeros(shape, np.float32)
    if ".norm" in name and name.endswith("weight") and len(shape) == 1:
        return (1.0 + n(0.1)).astype(np.float32)
    if ".gamma" in name:
        # LayerScale: trained DINOv2 gammas are O(0.1..1); keeps 24 residual adds well-conditioned.
        return (0.25 + 0.1 * g.random(shape, dtype=np.float32)).astype(np.float32)
    if name.endswith("output_conv2.2.bias"):
        # default torch init lands < 0 and the trailing ReLU zeroes the depth map
        # (SURVEY.md section 8 a-4); keep the output strictly inside the ReLU's active range.
        return np.full(shape, 1.0, np.float32)
    if name.endswith("output_conv2.2.weight"):
        return (0.05 + n(0.1)).astype(np.float32)     # mean-positive: most pixels stay above the ReLU
    if name.endswith("bias"):
        return n(0.1)
    return n(1.0 / np.sqrt(_fan_in(name, shape)))


def depth_anything_weights(cfg: DepthCfg | str = "vitl", seed: int = 1234) -> Dict[str, np.ndarray]:
    """name -> float32 ndarray, reference state_dict naming."""
    if isinstance(cfg, str):
        cfg = DEPTH_CFGS[cfg]
    return {name: _depth_tensor(seed, name, shape) for name, shape in depth_param_shapes(cfg)}


# ----------------------------------------------------------------------------
# Synthetic frames
# ----------------------------------------------------------------------------
def frames(n: int, height: int, width: int, seed: int = 0) -> np.ndarray:
    """n RGB uint8 frames [n, H, W, 3] = smooth seeded blobs + noise.

    Pure white noise gives the ViT nothing spatially coherent to respond to; a
    low-frequency texture plus noise gives depth maps with a healthy range.
    """
    g = np.random.default_rng([seed, 0xF4A3E5])
    out = np.empty((n, height, width, 3), np.uint8)
    yy = np.linspace(0.0, 1.0, height, dtype=np.float32)[:, None]
    xx = np.linspace(0.0, 1.0, width, dtype=np.float32)[None, :]
    for i in range(n):
        img = np.zeros((height, width, 3), np.float32)
        for _ in range(6):
            fx, fy = g.uniform(0.5, 6.0, 2)
            ph = g.uniform(0, 2 * np.pi, 3)
            amp = g.uniform(0.2, 1.0, 3)
            for c in range(3):
                img[..., c] += amp[c] * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph[c])
        img = (img - img.min()) / (img.max() - img.min() + 1e-6)
        noise = g.integers(0, 64, (height, width, 3), dtype=np.int32)
        out[i] = np.clip(img * 191.0 + noise, 0, 255).astype(np.uint8)
    return out


# ----------------------------------------------------------------------------
# RAFT (basic, hdim = cdim = 128, 4 levels, radius 4): /root/reference/bands/raft/raft.py:24-58,
# extractor.py:118-192, update.py:79-136.  Checkpoint naming without the `module.` prefix that
# bands/flow_raft.py:42-44 strips.
# ----------------------------------------------------------------------------
def raft_param_shapes() -> List[Tuple[str, Tuple[int, ...]]]:
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def conv(name, co, ci, kh, kw):
        out.append((name + ".weight", (co, ci, kh, kw)))
        out.append((name + ".bias", (co,)))

    def bn(name, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            out.append((f"{name}.{s}", (c,)))
        out.append((name + ".num_batches_tracked", ()))

    for enc, has_bn in (("fnet", False), ("cnet", True)):
        if has_bn:
            bn(enc + ".norm1", 64)
        conv(enc + ".conv1", 64, 3, 7, 7)
        cin = 64
        for li, (dim, stride) in enumerate(((64, 1), (96, 2), (128, 2)), start=1):
            for bi in range(2):
                p = f"{enc}.layer{li}.{bi}"
                ci = cin if bi == 0 else dim
                conv(p + ".conv1", dim, ci, 3, 3)
                conv(p + ".conv2", dim, dim, 3, 3)
                if has_bn:
                    bn(p + ".norm1", dim)
                    bn(p + ".norm2", dim)
                    if bi == 0 and stride != 1:
                        bn(p + ".norm3", dim)
                if bi == 0 and stride != 1:
                    conv(p + ".downsample.0", dim, ci, 1, 1)
                    if has_bn:
                        bn(p + ".downsample.1", dim)
            cin = dim
        conv(enc + ".conv2", 256, 128, 1, 1)
    u = "update_block."
    conv(u + "encoder.convc1", 256, 324, 1, 1)
    conv(u + "encoder.convc2", 192, 256, 3, 3)
    conv(u + "encoder.convf1", 128, 2, 7, 7)
    conv(u + "encoder.convf2", 64, 128, 3, 3)
    conv(u + "encoder.conv", 126, 256, 3, 3)
    for g in ("z", "r", "q"):
        conv(u + f"gru.conv{g}1", 128, 384, 1, 5)
        conv(u + f"gru.conv{g}2", 128, 384, 5, 1)
    conv(u + "flow_head.conv1", 256, 128, 3, 3)
    conv(u + "flow_head.conv2", 2, 256, 3, 3)
    conv(u + "mask.0", 256, 128, 3, 3)
    conv(u + "mask.2", 576, 256, 1, 1)
    return out


def raft_weights(seed: int = 4321) -> Dict[str, np.ndarray]:
    """name -> ndarray (float32; int64 scalar for num_batches_tracked), reference state_dict naming.

    `layerN.0.norm3.*` and `layerN.0.downsample.1.*` are the same module in the reference
    (extractor.py:32-44), so they get identical tensors.
    """
    w: Dict[str, np.ndarray] = {}
    for name, shape in raft_param_shapes():
        key = name.replace(".downsample.1.", ".norm3.")
        g = _rng(seed, key)
        if name.endswith("num_batches_tracked"):
            w[name] = np.array(1, np.int64)
        elif name.endswith("running_var"):
            w[name] = (0.5 + g.random(shape, dtype=np.float32)).astype(np.float32)
        elif name.endswith("running_mean"):
            w[name] = (g.standard_normal(shape, dtype=np.float32) * np.float32(0.1))
        elif len(shape) == 1 and (".norm" in name or ".downsample.1." in name):
            if name.endswith("weight"):
                w[name] = (1.0 + 0.1 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
            else:
                w[name] = (0.1 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif name.endswith("bias"):
            w[name] = (0.05 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 1.4                                   # ~He init keeps ReLU stacks at unit scale
            if "flow_head.conv2" in name:
                gain = 0.2                               # ~1 px (1/8 res) updates per iteration: contractive recurrence
            elif ".gru." in name or "mask.2" in name or name.endswith("conv2.weight") and name.count(".") == 2:
                gain = 1.0
            w[name] = (g.standard_normal(shape, dtype=np.float32) * np.float32(gain / np.sqrt(fan_in)))
    return w


def frame_pair_sequence(n: int, height: int, width: int, seed: int = 0, shift: Tuple[float, float] = (2.5, -1.5)) -> np.ndarray:
    """n frames of a smooth seeded texture translated by `shift` pixels per frame (dx, dy): flow is non-degenerate."""
    g = np.random.default_rng([seed, 0xF10])
    H, W = height + 64, width + 64
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    tex = np.zeros((H, W, 3), np.float32)
    for _ in range(24):
        fx, fy = g.uniform(0.01, 0.25, 2)
        ph, amp = g.uniform(0, 2 * np.pi, 3), g.uniform(0.2, 1.0, 3)
        for c in range(3):
            tex[..., c] += amp[c] * np.sin(fx * xx + fy * yy * (1 if c != 1 else -1) + ph[c])
    tex = (tex - tex.min()) / (tex.max() - tex.min())
    out = np.empty((n, height, width, 3), np.uint8)
    from scipy.ndimage import map_coordinates
    for i in range(n):
        ys = yy[:height, :width] + 32 + shift[1] * i
        xs = xx[:height, :width] + 32 + shift[0] * i
        for c in range(3):
            out[i, ..., c] = np.clip(map_coordinates(tex[..., c], [ys, xs], order=1) * 255.0, 0, 255).astype(np.uint8)
    return out


# ----------------------------------------------------------------------------
# SOLOv2 (ResNet + FPN + SOLOV2Head), mmdet 2.x state_dict naming as loaded by
# /root/reference/bands/mask_mmdet.py:36-39 (init_detector): backbone.* (models/backbones/resnet.py:306-659),
# neck.* (models/necks/fpn.py:11-204), mask_head.* (models/dense_heads/solov2_head.py:19-292).
# ----------------------------------------------------------------------------
@dataclass(frozen=True)
class MaskCfg:
    blocks: Tuple[int, int, int, int] = (3, 4, 23, 3)        # ResNet-101
    scale_long: int = 1333                                   # test pipeline img_scale=(1333, 800), keep_ratio
    scale_short: int = 800
    num_classes: int = 80
    feat_channels: int = 512                                 # head stacked convs
    stacked_convs: int = 4
    num_grids: Tuple[int, ...] = (40, 36, 24, 16, 12)
    strides: Tuple[int, ...] = (8, 8, 16, 32, 32)
    mask_feat_channels: int = 128
    mask_out_channels: int = 256
    nms_pre: int = 500
    score_thr: float = 0.1
    mask_thr: float = 0.5
    filter_thr: float = 0.05
    sigma: float = 2.0
    max_per_img: int = 100


MASK_CFGS = {
    "r101": MaskCfg(),
    "r50": MaskCfg(blocks=(3, 4, 6, 3)),
    # small configurations for CPU-sized parity cases (same code paths, fewer blocks / pixels)
    "tiny": MaskCfg(blocks=(1, 1, 1, 1), scale_long=320, scale_short=192, feat_channels=128,
                    num_grids=(12, 10, 8, 6, 4)),
    "r18ish": MaskCfg(blocks=(2, 2, 2, 2), scale_long=448, scale_short=256, feat_channels=256,
                      num_grids=(20, 18, 12, 8, 6)),
}

# COCO class order used by mmdet (model.CLASSES); the band keeps the 11 animate classes (mask_mmdet.py:30)
COCO_CLASSES = ('person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck', 'boat', 'traffic light',
                'fire hydrant', 'stop sign', 'parking meter', 'bench', 'bird', 'cat', 'dog', 'horse', 'sheep', 'cow',
                'elephant', 'bear', 'zebra', 'giraffe', 'backpack', 'umbrella', 'handbag', 'tie', 'suitcase', 'frisbee',
                'skis', 'snowboard', 'sports ball', 'kite', 'baseball bat', 'baseball glove', 'skateboard', 'surfboard',
                'tennis racket', 'bottle', 'wine glass', 'cup', 'fork', 'knife', 'spoon', 'bowl', 'banana', 'apple',
                'sandwich', 'orange', 'broccoli', 'carrot', 'hot dog', 'pizza', 'donut', 'cake', 'chair', 'couch',
                'potted plant', 'bed', 'dining table', 'toilet', 'tv', 'laptop', 'mouse', 'remote', 'keyboard',
                'cell phone', 'microwave', 'oven', 'toaster', 'sink', 'refrigerator', 'book', 'clock', 'vase',
                'scissors', 'teddy bear', 'hair drier', 'toothbrush')
BAND_CLASSES = ('person', 'bird', 'cat', 'dog', 'horse', 'sheep', 'cow', 'elephant', 'bear', 'zebra', 'giraffe')


def solov2_param_shapes(cfg: MaskCfg) -> List[Tuple[str, Tuple[int, ...]]]:
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def bn(name, c):
        for s in ("weight", "bias", "running_mean", "running_var"):
            out.append((f"{name}.{s}", (c,)))
        out.append((name + ".num_batches_tracked", ()))

    def gn(name, c):
        out.append((name + ".weight", (c,)))
        out.append((name + ".bias", (c,)))

    out.append(("backbone.conv1.weight", (64, 3, 7, 7)))
    bn("backbone.bn1", 64)
    inpl = 64
    for li, nb in enumerate(cfg.blocks, start=1):
        planes = 64 << (li - 1)
        for b in range(nb):
            p = f"backbone.layer{li}.{b}"
            out.append((p + ".conv1.weight", (planes, inpl, 1, 1)))
            bn(p + ".bn1", planes)
            out.append((p + ".conv2.weight", (planes, planes, 3, 3)))
            bn(p + ".bn2", planes)
            out.append((p + ".conv3.weight", (planes * 4, planes, 1, 1)))
            bn(p + ".bn3", planes * 4)
            if b == 0:
                out.append((p + ".downsample.0.weight", (planes * 4, inpl, 1, 1)))
                bn(p + ".downsample.1", planes * 4)
            inpl = planes * 4
    for i, c in enumerate((256, 512, 1024, 2048)):
        out.append((f"neck.lateral_convs.{i}.conv.weight", (256, c, 1, 1)))
        out.append((f"neck.lateral_convs.{i}.conv.bias", (256,)))
        out.append((f"neck.fpn_convs.{i}.conv.weight", (256, 256, 3, 3)))
        out.append((f"neck.fpn_convs.{i}.conv.bias", (256,)))
    mf, fc = cfg.mask_feat_channels, cfg.feat_channels
    h = "mask_head.mask_feature_head."
    for i in range(4):                                      # start_level 0 .. end_level 3
        for j in range(max(i, 1)):
            cin = 256 if j == 0 else mf
            if i == 3 and j == 0:
                cin = 258                                   # + (x, y) coordinate channels on the coarsest level
            out.append((f"{h}convs_all_levels.{i}.conv{j}.conv.weight", (mf, cin, 3, 3)))
            gn(f"{h}convs_all_levels.{i}.conv{j}.gn", mf)
    out.append((h + "conv_pred.conv.weight", (cfg.mask_out_channels, mf, 1, 1)))
    gn(h + "conv_pred.gn", cfg.mask_out_channels)
    for i in range(cfg.stacked_convs):
        out.append((f"mask_head.kernel_convs.{i}.conv.weight", (fc, 258 if i == 0 else fc, 3, 3)))
        gn(f"mask_head.kernel_convs.{i}.gn", fc)
        out.append((f"mask_head.cls_convs.{i}.conv.weight", (fc, 256 if i == 0 else fc, 3, 3)))
        gn(f"mask_head.cls_convs.{i}.gn", fc)
    out.append(("mask_head.conv_cls.weight", (cfg.num_classes, fc, 3, 3)))
    out.append(("mask_head.conv_cls.bias", (cfg.num_classes,)))
    out.append(("mask_head.conv_kernel.weight", (cfg.mask_out_channels, fc, 3, 3)))
    out.append(("mask_head.conv_kernel.bias", (cfg.mask_out_channels,)))
    return out


def solov2_weights(cfg: MaskCfg, seed: int = 777) -> Dict[str, np.ndarray]:
    """Seeded weights that keep every stage at unit scale and give a usable detector statistic: class logits with
    a negative bias so a few hundred grid cells clear score_thr, a handful of them in the band's classes above 0.5,
    and dynamic-conv logits with enough contrast that masks are clean blobs rather than threshold noise."""
    w: Dict[str, np.ndarray] = {}
    band_ids = [COCO_CLASSES.index(c) for c in BAND_CLASSES]
    for name, shape in solov2_param_shapes(cfg):
        g = _rng(seed, name)
        if name.endswith("num_batches_tracked"):
            w[name] = np.array(1, np.int64)
        elif name.endswith("running_var"):
            w[name] = (0.5 + g.random(shape, dtype=np.float32)).astype(np.float32)
        elif name.endswith("running_mean"):
            w[name] = g.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
        elif len(shape) == 1 and (".bn" in name or ".downsample.1." in name or ".gn." in name):
            if name.endswith("weight"):
                base = 0.4 if name.endswith(".bn3.weight") else 1.0      # keep the residual sum from growing per block
                w[name] = (base * (1.0 + 0.1 * g.standard_normal(shape, dtype=np.float32))).astype(np.float32)
            else:
                w[name] = (0.1 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif name == "mask_head.conv_cls.bias":
            # class logits come out ~N(bias, 1.55^2) (twice that spread for the band's classes, whose conv_cls rows are
            # doubled below): ~300 cells of the other classes clear score_thr, and the band's classes get a long tail -
            # few cells over score_thr, the best of them (about 4 sigma out) well over 0.5
            from scipy.stats import norm
            pts = float(sum(g_ * g_ for g_ in cfg.num_grids))
            n_band = sum(1 for i in band_ids if i < shape[0])
            b_other = float(np.log(cfg.score_thr / (1 - cfg.score_thr))) - 1.55 * norm.isf(min(0.4, 300.0 / (pts * max(shape[0] - n_band, 1))))
            # measured: wide heads (feat_channels 512) give smoother, lighter-tailed logit fields - their maximum sits
            # ~2.7 sigma out instead of ~4 - so they get a larger row gain and a matching bias
            b_band = 2.5 - (2.7 * 5.3 if cfg.feat_channels >= 512 else 4.0 * 3.1)
            b = np.full(shape, b_other, np.float32)
            b[[i for i in band_ids if i < shape[0]]] = b_band
            w[name] = b
        elif name.endswith("bias"):
            w[name] = (0.05 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 1.4
            if name == "mask_head.conv_cls.weight":
                gain = 2.2
            elif name == "mask_head.conv_kernel.weight":
                gain = 3.0                               # high-contrast dynamic-conv logits: crisp masks, maskness near 1
            elif ".downsample.0." in name or ".conv3." in name or "lateral_convs" in name or "fpn_convs" in name \
                    or "conv_pred" in name:
                gain = 1.0
            w[name] = g.standard_normal(shape, dtype=np.float32) * np.float32(gain / np.sqrt(fan_in))
            if name == "mask_head.conv_cls.weight":
                w[name][[i for i in band_ids if i < shape[0]]] *= np.float32(3.0 if cfg.feat_channels >= 512 else 2.0)
    return w


# ----------------------------------------------------------------------------
# ZoeDepth metric head over the Depth-Anything ViT-L + DPT core (`depth_anything --metric indoor|outdoor`):
# /root/reference/bands/patchfusion/zoedepth/models/zoedepth/zoedepth_v1.py:39-137, config_zoedepth.json
# (n_bins 64, bin_embedding_dim 128, softplus bin centres, n_attractors [16, 8, 4, 1], alpha 1000 in the config but 300
# in effect - the layer calls inv_attractor with its defaults -, gamma 2, kind mean, type inv, min_temp 0.0212, max_temp 50, img_size [392, 518]).  state_dict naming of ZoeDepth: the core's
# tensors sit under `core.core.` (DepthAnythingCore.core = DPT_DINOv2).
# ----------------------------------------------------------------------------
ZOE = dict(n_bins=64, bin_dim=128, n_attractors=(16, 8, 4, 1), alpha=300.0, gamma=2, min_temp=0.0212, max_temp=50.0,
           img_size=(392, 518), features=256, last_in=33)


def zoe_head_param_shapes() -> List[Tuple[str, Tuple[int, ...]]]:
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def conv(name, co, ci):
        out.append((name + ".weight", (co, ci, 1, 1)))
        out.append((name + ".bias", (co,)))

    F_, nb, bd = ZOE["features"], ZOE["n_bins"], ZOE["bin_dim"]
    conv("conv2", F_, F_)
    conv("seed_bin_regressor._net.0", 256, F_)
    conv("seed_bin_regressor._net.2", nb, 256)
    conv("seed_projector._net.0", 128, F_)
    conv("seed_projector._net.2", bd, 128)
    for i, na in enumerate(ZOE["n_attractors"]):
        conv(f"projectors.{i}._net.0", 128, F_)
        conv(f"projectors.{i}._net.2", bd, 128)
        conv(f"attractors.{i}._net.0", 128, bd)
        conv(f"attractors.{i}._net.2", na, 128)
    bott = (ZOE["last_in"] + bd) // 2
    conv("conditional_log_binomial.mlp.0", bott, ZOE["last_in"] + bd)
    conv("conditional_log_binomial.mlp.2", 4, bott)
    return out


def zoe_weights(seed: int = 2468) -> Dict[str, np.ndarray]:
    """ZoeDepth state dict on the seeded ViT-L core: `core.core.<depth-anything name>` + the metric head.  The seed
    regressor's bias spreads the 64 softplus bin centres over ~0.5 .. 8 (metres), attractor outputs land in the same
    range, so the attractors move bins and the log-binomial picks among distinct depths."""
    core = depth_anything_weights("vitl", seed=1234)
    w: Dict[str, np.ndarray] = {"core.core." + k: v for k, v in core.items()}
    for name, shape in zoe_head_param_shapes():
        g = _rng(seed, name)
        if name.endswith(".bias"):
            b = (0.1 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
            if name == "seed_bin_regressor._net.2.bias":
                b = np.linspace(0.5, 8.0, shape[0]).astype(np.float32)
            elif name.startswith("attractors.") and name.endswith("_net.2.bias"):
                b = np.linspace(1.0, 7.0, shape[0]).astype(np.float32) if shape[0] > 1 else np.array([3.0], np.float32)
            w[name] = b
        else:
            gain = 1.4 if name.endswith("_net.0.weight") or name.endswith("mlp.0.weight") else 1.0
            if name == "seed_bin_regressor._net.2.weight" or (name.startswith("attractors.") and name.endswith("_net.2.weight")):
                gain = 0.5
            w[name] = g.standard_normal(shape, dtype=np.float32) * np.float32(gain / np.sqrt(shape[1]))
    return w


# ---------------------------------------------------------------------------
# GMFlow (bands/flow_gmflow.py defaults: feature_channels 128, 1 scale, 1 head, swin attention with 2 x 2 splits, 6 blocks, ffn x 4)
# ---------------------------------------------------------------------------
def gmflow_param_shapes(channels: int = 128, layers: int = 6, ffn: int = 4, upsample: int = 8):
    """(name, shape) of every tensor of the reference's GMFlow state_dict at the band's defaults (bands/gmflow/gmflow.py:12-47,
    backbone.py:5-117, transformer.py:104-139, 284-298).  InstanceNorm2d layers are affine-free: no entries."""
    C = channels
    out = [("backbone.conv1.weight", (64, 3, 7, 7))]
    cin = 64
    for li, dim in enumerate((64, 96, 128), start=1):
        for bi in range(2):
            p = f"backbone.layer{li}.{bi}."
            out += [(p + "conv1.weight", (dim, cin if bi == 0 else dim, 3, 3)), (p + "conv2.weight", (dim, dim, 3, 3))]
            if bi == 0 and li > 1:
                out += [(p + "downsample.0.weight", (dim, cin, 1, 1)), (p + "downsample.0.bias", (dim,))]
        cin = dim
    out += [("backbone.conv2.weight", (C, 128, 1, 1)), ("backbone.conv2.bias", (C,))]
    for i in range(layers):
        for part, has_ffn in (("self_attn", False), ("cross_attn_ffn", True)):
            p = f"transformer.layers.{i}.{part}."
            out += [(p + n + ".weight", (C, C)) for n in ("q_proj", "k_proj", "v_proj", "merge")]
            out += [(p + "norm1.weight", (C,)), (p + "norm1.bias", (C,))]
            if has_ffn:
                out += [(p + "mlp.0.weight", (2 * C * ffn, 2 * C)), (p + "mlp.2.weight", (C, 2 * C * ffn)),
                        (p + "norm2.weight", (C,)), (p + "norm2.bias", (C,))]
    out += [("feature_flow_attn.q_proj.weight", (C, C)), ("feature_flow_attn.q_proj.bias", (C,)),
            ("feature_flow_attn.k_proj.weight", (C, C)), ("feature_flow_attn.k_proj.bias", (C,)),
            ("upsampler.0.weight", (256, 2 + C, 3, 3)), ("upsampler.0.bias", (256,)),
            ("upsampler.2.weight", (upsample * upsample * 9, 256, 1, 1)), ("upsampler.2.bias", (upsample * upsample * 9,))]
    return out


def gmflow_weights(seed: int = 2468) -> Dict[str, np.ndarray]:
    """Seeded float32 tensors under the reference's state_dict names.  Scales keep the network in the regime of a trained one: unit-scale
    features out of the backbone, LayerNorm'd messages at ~0.3 of the stream they are added to, so the matching softmax is sharply peaked on
    the true correspondence of the translated synthetic texture without being a hard argmax."""
    w: Dict[str, np.ndarray] = {}
    for name, shape in gmflow_param_shapes():
        g = _rng(seed, name)
        if ".norm" in name:
            w[name] = ((0.3 + 0.03 * g.standard_normal(shape, dtype=np.float32)) if name.endswith("weight")
                       else 0.03 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        elif name.endswith("bias"):
            w[name] = (0.05 * g.standard_normal(shape, dtype=np.float32)).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = 1.4 if name.startswith("backbone.") and "conv2.weight" != name.split("backbone.")[-1] or name.startswith("upsampler.0") else 1.0
            if name.startswith("upsampler.2"):
                gain = 2.0                                # convex-combination logits with some contrast
            w[name] = (g.standard_normal(shape, dtype=np.float32) * np.float32(gain / np.sqrt(fan_in)))
    return w


# ----------------------------------------------------------------------------
# Heavy-tailed variants (round 3): trained checkpoints have outlier channels - a few LayerNorm gains tens of times the rest, a few
# convolution channels far above unit scale - which the unit-scale synthetic weights never exercise.  These variants re-scale the SAME
# seeded tensors: a producer's channel c is multiplied by `up` and every consumer's weights on that input channel divided by `down`
# (< up, so the channel also carries `up / down` times its share of the consumer's output).  The functions stay well conditioned (the
# reference model is still the arbiter: oracle/make_golden.py heavy) while the activations the split-precision copies have to hold
# reach tens to hundreds - what the fixed power-of-two e4m3 scales of the engine must survive (tests/test_gpu_outliers.py).
# ----------------------------------------------------------------------------
def _heavy_channels(seed: int, name: str, n: int, k: int) -> np.ndarray:
    return np.sort(_rng(seed, name + "#heavy").choice(n, size=k, replace=False))


def depth_anything_weights_heavy(cfg: DepthCfg | str = "vitl", seed: int = 1234, up: float = 50.0, down: float = 12.5,
                                 conv_up: float = 30.0, conv_down: float = 7.5) -> Dict[str, np.ndarray]:
    """depth_anything_weights with 4 outlier channels per LayerNorm (gamma and beta x `up`, the consuming qkv / fc1 / projects columns /
    `down`) and 3 outlier channels per DPT `layerN_rn` output (x `conv_up`; the refinenet's consumers of that channel / `conv_down`)."""
    if isinstance(cfg, str):
        cfg = DEPTH_CFGS[cfg]
    w = {k: v.copy() for k, v in depth_anything_weights(cfg, seed).items()}
    D = cfg.embed_dim
    for i in range(cfg.depth):
        p = f"pretrained.blocks.{i}."
        for norm, consumer in (("norm1", "attn.qkv.weight"), ("norm2", "mlp.fc1.weight")):
            ch = _heavy_channels(seed, p + norm, D, 4)
            w[p + norm + ".weight"][ch] *= np.float32(up)
            w[p + norm + ".bias"][ch] *= np.float32(up)
            w[p + consumer][:, ch] /= np.float32(down)
    ch = _heavy_channels(seed, "pretrained.norm", D, 4)          # the final norm feeds the four DPT taps
    w["pretrained.norm.weight"][ch] *= np.float32(up)
    w["pretrained.norm.bias"][ch] *= np.float32(up)
    for i in range(4):
        w[f"depth_head.projects.{i}.weight"][:, ch] /= np.float32(down)
    F = cfg.features
    for i in range(4):
        ch = _heavy_channels(seed, f"depth_head.scratch.layer{i + 1}_rn", F, 3)
        w[f"depth_head.scratch.layer{i + 1}_rn.weight"][ch] *= np.float32(conv_up)
        r = f"depth_head.scratch.refinenet{i + 1}."
        for cons in ("resConfUnit1.conv1.weight", "resConfUnit2.conv1.weight", "out_conv.weight"):
            w[r + cons][:, ch] /= np.float32(conv_down)
    return w


def raft_weights_heavy(seed: int = 4321, up: float = 30.0, down: float = 7.5) -> Dict[str, np.ndarray]:
    """raft_weights with 3 outlier channels in every cnet residual block's first norm (eval BatchNorm: weight and bias x `up`, so the
    ReLU'd map that conv2 reads carries them; conv2's weights on those inputs / `down`) and in fnet's / cnet's last 1x1 conv input
    (layer3.1's second norm cannot be re-scaled under InstanceNorm, so there the 1x1 `conv2` rows are scaled instead: output channels
    x 4 on the correlation features / context)."""
    w = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in raft_weights(seed).items()}
    dims = (64, 96, 128)
    for li in range(3):
        for bi in range(2):
            p = f"cnet.layer{li + 1}.{bi}."
            ch = _heavy_channels(seed, p + "norm1", dims[li], 3)
            w[p + "norm1.weight"][ch] *= np.float32(up)
            w[p + "norm1.bias"][ch] *= np.float32(up)
            w[p + "conv2.weight"][:, ch] /= np.float32(down)
    for enc in ("fnet", "cnet"):
        ch = _heavy_channels(seed, enc + ".conv2", 128, 3)
        w[enc + ".conv2.weight"][:, ch] *= np.float32(4.0)
    return w


# ----------------------------------------------------------------------------
# Weight cache for multi-rank launches: the ranks of one node all need the same seeded tensors, and generating ViT-L's 335 M parameters
# takes ~10 s of one core.  bench.py lets rank 0 generate and store them (uncompressed .npy files in a tmpfs directory keyed by generator,
# arguments and this file's hash), waits on a barrier and has the other ranks memory-map the files.
# ----------------------------------------------------------------------------
def _cache_dir(kind: str, key: str) -> str:
    import hashlib
    import os
    src = hashlib.sha1(open(os.path.abspath(__file__), "rb").read()).hexdigest()[:12]
    base = os.environ.get("PRISMA_SYNTH_CACHE") or ("/dev/shm/prisma_synth_cache" if os.path.isdir("/dev/shm") else os.path.join(os.path.expanduser("~"), ".cache", "prisma_synth"))
    return os.path.join(base, f"{kind}-{key}-{src}")


def cached_weights(kind: str, *args) -> Dict[str, np.ndarray]:
    """{"depth": depth_anything_weights, "raft": raft_weights, "gmflow": gmflow_weights, "solov2": solov2_weights}[kind](*args), through
    the cache.  A complete cache entry has a `done` marker written last; an incomplete one is regenerated privately (no locks needed)."""
    import os
    gen = {"depth": depth_anything_weights, "raft": raft_weights, "gmflow": gmflow_weights, "solov2": solov2_weights}[kind]
    key = "_".join(str(getattr(a, "name", a)) for a in args) or "default"
    d = _cache_dir(kind, key)
    if os.path.exists(os.path.join(d, "done")):
        names = [l.rstrip("\n") for l in open(os.path.join(d, "names.txt"))]
        return {n: np.load(os.path.join(d, "%04d.npy" % i), mmap_mode="r") for i, n in enumerate(names)}
    w = gen(*args)
    try:
        tmp = d + ".tmp.%d" % os.getpid()
        os.makedirs(tmp, exist_ok=True)
        for i, (n, v) in enumerate(w.items()):
            np.save(os.path.join(tmp, "%04d.npy" % i), np.asarray(v))
        with open(os.path.join(tmp, "names.txt"), "w") as f:
            f.writelines(n + "\n" for n in w)
        open(os.path.join(tmp, "done"), "w").close()
        os.replace(tmp, d)                  # atomic publish; if another process won the race the rename fails and its entry is used next time
    except OSError:
        pass
    return w
