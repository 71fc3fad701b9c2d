"""ORACLE (test infrastructure, never the product path) for the flow_raft band.

CPU restatement in numpy + torch.nn.functional fp32 of /root/reference/bands/flow_raft.py and the
RAFT model it drives.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it, as the checker.  Pinned against the imported reference modules by oracle/make_golden.py
(vectors in tests/golden/raft_*.npz).

Parity status
  * RAFT.forward (encoders, all-pairs correlation + pyramid, 9x9x4 lookup, SepConvGRU update, convex
    upsample), InputPadder, process_flow: PINNED.
  * cv2.resize(frame, fx=fy=0.75, INTER_CUBIC) on uint8 (bands/flow_raft.py:100): PARITY UNPINNED -
    opencv-python 4.8.1.78 is absent; `cv_resize_cubic_u8` restates OpenCV's published 8-bit path
    (11-bit fixed-point coefficients, a = -0.75, replicated border, rounding cast after 22 bits).

Paths cited are relative to /root/reference/.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .depth_oracle import _cubic_coeffs, hue_to_rgb


def _t(w, k):
    return torch.from_numpy(np.ascontiguousarray(w[k]))


# ---------------------------------------------------------------------------
# frame prep (bands/flow_raft.py:99-101, bands/common/flow.py:13-16, 43-61)
# ---------------------------------------------------------------------------
def scaled_size(h: int, w: int, scale: float) -> Tuple[int, int]:
    """cv2.resize(..., fx=fy=scale): dsize = round-half-even(src * scale)."""
    return int(np.rint(h * scale)), int(np.rint(w * scale))


def cubic_taps_u8(src: int, dst: int, scale: float):
    """4 clamped source indices and 4 int16 coefficients (x2048) per destination index."""
    inv = 1.0 / scale
    d = np.arange(dst, dtype=np.float64)
    fx = ((d + 0.5) * inv - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = fx - sx.astype(np.float32)
    idx = np.clip(sx[:, None] + np.arange(-1, 3)[None, :], 0, src - 1)
    co = np.rint(_cubic_coeffs(fx).astype(np.float64) * 2048.0).astype(np.int32)     # saturate_cast<short>(c * 2048)
    return idx, co


def cv_resize_cubic_u8(img: np.ndarray, scale: float) -> np.ndarray:
    h, w = img.shape[:2]
    oh, ow = scaled_size(h, w, scale)
    xi, xc = cubic_taps_u8(w, ow, scale)
    yi, yc = cubic_taps_u8(h, oh, scale)
    s = img.astype(np.int64)
    tmp = np.zeros((h, ow, img.shape[2]), np.int64)
    for t in range(4):
        tmp += s[:, xi[:, t], :] * xc[:, t].astype(np.int64)[None, :, None]
    out = np.zeros((oh, ow, img.shape[2]), np.int64)
    for t in range(4):
        out += tmp[yi[:, t], :, :] * yc[:, t].astype(np.int64)[:, None, None]
    out = (out + (1 << 21)) >> 22
    return np.clip(out, 0, 255).astype(np.uint8)


def pad_amounts(h: int, w: int, factor: int = 8):
    """InputPadder('sintel'): [left, right, top, bottom] (bands/common/flow.py:43-55)."""
    ph = (((h // factor) + 1) * factor - h) % factor
    pw = (((w // factor) + 1) * factor - w) % factor
    return [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]


# ---------------------------------------------------------------------------
# RAFT (bands/raft/*.py)
# ---------------------------------------------------------------------------
def _norm(w, p: str, x: torch.Tensor, kind: str) -> torch.Tensor:
    if kind == "instance":                                   # nn.InstanceNorm2d defaults: no affine, eps 1e-5
        return F.instance_norm(x, eps=1e-5)
    return F.batch_norm(x, _t(w, p + ".running_mean"), _t(w, p + ".running_var"), _t(w, p + ".weight"),
                        _t(w, p + ".bias"), training=False, eps=1e-5)


def _conv(w, p: str, x, stride=1, padding=0):
    return F.conv2d(x, _t(w, p + ".weight"), _t(w, p + ".bias"), stride=stride, padding=padding)


def _resblock(w, p: str, x, kind: str, stride: int):
    """ResidualBlock.forward (raft/extractor.py:46-56)."""
    y = F.relu(_norm(w, p + ".norm1", _conv(w, p + ".conv1", x, stride, 1), kind))
    y = F.relu(_norm(w, p + ".norm2", _conv(w, p + ".conv2", y, 1, 1), kind))
    if stride != 1:
        x = _norm(w, p + ".norm3", _conv(w, p + ".downsample.0", x, stride, 0), kind)
    return F.relu(x + y)


def encoder(w, enc: str, x: torch.Tensor, kind: str) -> torch.Tensor:
    """BasicEncoder.forward (raft/extractor.py:171-192)."""
    x = F.relu(_norm(w, enc + ".norm1", _conv(w, enc + ".conv1", x, 2, 3), kind))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        x = _resblock(w, f"{enc}.layer{li}.0", x, kind, stride)
        x = _resblock(w, f"{enc}.layer{li}.1", x, kind, 1)
    return _conv(w, enc + ".conv2", x)


def corr_pyramid(fmap1, fmap2, levels=4):
    """CorrBlock.__init__ / corr (raft/corr.py:13-27, 52-60)."""
    b, d, h, wd = fmap1.shape
    c = torch.matmul(fmap1.view(b, d, h * wd).transpose(1, 2), fmap2.view(b, d, h * wd))
    c = c.view(b * h * wd, 1, h, wd) / torch.sqrt(torch.tensor(d).float())
    pyr = [c]
    for _ in range(levels - 1):
        c = F.avg_pool2d(c, 2, stride=2)
        pyr.append(c)
    return pyr


def corr_lookup(pyr, coords, r=4):
    """CorrBlock.__call__ (raft/corr.py:29-50) + bilinear_sampler (raft/utils/utils.py:58-72).
    Channel k = level*81 + i*9 + j samples x + (i - r), y + (j - r): the first window index moves x."""
    b, _, h, wd = coords.shape
    co = coords.permute(0, 2, 3, 1).reshape(b * h * wd, 1, 1, 2)
    d = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)
    out = []
    for i, c in enumerate(pyr):
        g = co / 2 ** i + delta
        H, W = c.shape[-2:]
        gx = 2 * g[..., 0:1] / (W - 1) - 1
        gy = 2 * g[..., 1:2] / (H - 1) - 1
        s = F.grid_sample(c, torch.cat([gx, gy], -1), align_corners=True)
        out.append(s.view(b, h, wd, -1))
    return torch.cat(out, -1).permute(0, 3, 1, 2).contiguous().float()


def _gru_half(w, p: str, suffix: str, h, x, pad):
    hx = torch.cat([h, x], 1)
    z = torch.sigmoid(_conv(w, f"{p}.convz{suffix}", hx, 1, pad))
    r = torch.sigmoid(_conv(w, f"{p}.convr{suffix}", hx, 1, pad))
    q = torch.tanh(_conv(w, f"{p}.convq{suffix}", torch.cat([r * h, x], 1), 1, pad))
    return (1 - z) * h + z * q


def update_block(w, net, inp, corr, flow, want_mask: bool):
    """BasicUpdateBlock.forward (raft/update.py:79-97, 33-60, 6-14, 122-136)."""
    u = "update_block."
    cor = F.relu(_conv(w, u + "encoder.convc1", corr))
    cor = F.relu(_conv(w, u + "encoder.convc2", cor, 1, 1))
    flo = F.relu(_conv(w, u + "encoder.convf1", flow, 1, 3))
    flo = F.relu(_conv(w, u + "encoder.convf2", flo, 1, 1))
    out = F.relu(_conv(w, u + "encoder.conv", torch.cat([cor, flo], 1), 1, 1))
    x = torch.cat([inp, out, flow], 1)
    net = _gru_half(w, u + "gru", "1", net, x, (0, 2))
    net = _gru_half(w, u + "gru", "2", net, x, (2, 0))
    delta = _conv(w, u + "flow_head.conv2", F.relu(_conv(w, u + "flow_head.conv1", net, 1, 1)), 1, 1)
    mask = None
    if want_mask:
        mask = 0.25 * _conv(w, u + "mask.2", F.relu(_conv(w, u + "mask.0", net, 1, 1)))
    return net, mask, delta


def upsample_flow(flow, mask):
    """RAFT.upsample_flow (raft/raft.py:73-84)."""
    n, _, h, wd = flow.shape
    mask = torch.softmax(mask.view(n, 1, 9, 8, 8, h, wd), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(n, 2, 9, 1, 1, h, wd)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(n, 2, 8 * h, 8 * wd)


def raft_forward(w: Dict[str, np.ndarray], image1: np.ndarray, image2: np.ndarray, iters: int = 12,
                 return_stages: bool = False):
    """RAFT.forward(test_mode=True) (raft/raft.py:87-146): images float32 [B,3,H,W] in 0..255, H,W % 8 == 0.
    The mask head only matters on the last iteration in test mode, so it is evaluated once."""
    st = {}
    with torch.no_grad():
        i1 = (2 * (torch.from_numpy(image1) / 255.0) - 1.0).contiguous()
        i2 = (2 * (torch.from_numpy(image2) / 255.0) - 1.0).contiguous()
        b = i1.shape[0]
        f = encoder(w, "fnet", torch.cat([i1, i2], 0), "instance").float()
        f1, f2 = f[:b], f[b:]
        pyr = corr_pyramid(f1, f2)
        c = encoder(w, "cnet", i1, "batch")
        net, inp = torch.tanh(c[:, :128]), torch.relu(c[:, 128:])
        h8, w8 = i1.shape[2] // 8, i1.shape[3] // 8
        ys, xs = torch.meshgrid(torch.arange(h8), torch.arange(w8), indexing="ij")
        coords0 = torch.stack([xs, ys], 0).float()[None].repeat(b, 1, 1, 1)
        coords1 = coords0.clone()
        st.update(fmap1=f1, fmap2=f2, net0=net, inp=inp)
        mask = None
        for it in range(iters):
            corr = corr_lookup(pyr, coords1)
            if it == 0:
                st["corr0"] = corr
            net, mask, delta = update_block(w, net, inp, corr, coords1 - coords0, want_mask=(it == iters - 1))
            coords1 = coords1 + delta
            if it in (0, iters - 1):
                st[f"flow_it{it}"] = coords1 - coords0
                st[f"net_it{it}"] = net
        flow_lo = coords1 - coords0
        flow_up = upsample_flow(flow_lo, mask)
    if return_stages:
        return flow_lo.numpy(), flow_up.numpy(), {k: v.numpy() for k, v in st.items()}
    return flow_lo.numpy(), flow_up.numpy()


def infer_pair(w, prev_u8: np.ndarray, curr_u8: np.ndarray, scale: float = 0.75, iters: int = 12, backward: bool = True):
    """bands/flow_raft.py:99-107 + infer (:51-62): two uint8 frames -> (fwd, bwd) float32 [H', W', 2] at the
    scaled resolution.  Index 0 of the reference's batch is prev->curr, index 1 curr->prev.
    backward=False evaluates the forward direction only (batch of 1; bwd is None) - what the engine computes without
    --backwards / --mask, and what bench.py's cpu_baseline times beside it."""
    a = cv_resize_cubic_u8(prev_u8, scale) if scale != 1.0 else prev_u8
    c = cv_resize_cubic_u8(curr_u8, scale) if scale != 1.0 else curr_u8
    ta = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1).float()[None]
    tc = torch.from_numpy(np.ascontiguousarray(c)).permute(2, 0, 1).float()[None]
    i1, i2 = (torch.cat([ta, tc], 0), torch.cat([tc, ta], 0)) if backward else (ta, tc)
    pad = pad_amounts(i1.shape[2], i1.shape[3])
    i1p, i2p = F.pad(i1, pad, mode="replicate"), F.pad(i2, pad, mode="replicate")
    _, up = raft_forward(w, i1p.numpy(), i2p.numpy(), iters)
    H, W = up.shape[2:]
    up = up[:, :, pad[2]:H - pad[3], pad[0]:W - pad[1]]
    return np.ascontiguousarray(up[0].transpose(1, 2, 0)), (np.ascontiguousarray(up[1].transpose(1, 2, 0)) if backward else None)


_ATAN_C = np.array([0.0, 0.24497866312686414, 0.4636476090008061, 0.6435011087932844, 0.7853981633974483])
_ATAN_COEF = [(-1.0) ** i / (2 * i + 1) for i in range(9)]


def atan2_rn(y: np.ndarray, x: np.ndarray) -> np.ndarray:
    """arctan2 in float64 from IEEE +, -, *, / only - the operation sequence of csrc/raft_kernels.hip atan2_rn, so the HIP
    kernel reproduces it bit for bit.  |error| < 5e-16: its float32 rounding is the correctly rounded float32 arctan2
    (tests/test_raft_oracle.py checks it against np.arctan2 in float64 and long double)."""
    y, x = np.asarray(y, np.float64), np.asarray(x, np.float64)
    ax, ay = np.abs(x), np.abs(y)
    hi, lo = np.maximum(ax, ay), np.minimum(ax, ay)
    with np.errstate(invalid="ignore", divide="ignore"):
        t = np.where(hi == 0.0, 0.0, lo / hi)
        k = np.floor(t * 4.0 + 0.5)
        c = k * 0.25
        u = (t - c) / (1.0 + t * c)
        u2 = u * u
        p = np.full_like(u, _ATAN_COEF[8])
        for i in range(7, -1, -1):
            p = p * u2 + _ATAN_COEF[i]
        r = _ATAN_C[np.where(np.isnan(k), 0, k).astype(np.int64)] + u * p
        r = np.where(ay > ax, 1.5707963267948966 - r, r)
        r = np.where(x < 0.0, 3.141592653589793 - r, r)
        r = np.where(y < 0.0, -r, r)
    return np.where(np.isnan(t), np.nan, r)


def process_flow(flow: np.ndarray, exact_atan2: bool = False):
    """bands/common/encode.py:98-126 (+ hue_to_rgb :13-28, saturation :73-78): polar HSV-style encode.
    dtypes follow numpy's promotion in the reference: distances / angle / hue*6 in the flow's float32,
    the colour ramp and the saturation blend in float64; uint8 conversion truncates.
    exact_atan2: np.arctan2 on float32 is host dependent (SVML, <= 4 ULP, on AVX512 builds; libm elsewhere - 38 % of random
    inputs differ in the last bit between the two), so the reference's bytes are only defined up to that.  False follows
    numpy on this host (bit-identical to the reference run on the same host: tests/golden/encode.npz); True is the
    correctly rounded float32 arctan2 (atan2_rn in double, rounded once) - the definition the HIP kernel implements."""
    flow = np.asarray(flow, np.float32)
    dist = np.sqrt(np.square(flow[..., 0]) + np.square(flow[..., 1]))
    mx = dist.max()
    with np.errstate(invalid="ignore", divide="ignore"):
        dx = flow[..., 0] / float(mx)
        dy = flow[..., 1] / float(mx)
        rad = np.sqrt(np.square(dx) + np.square(dy))
        at = atan2_rn(dy, dx).astype(np.float32) if exact_atan2 else np.arctan2(dy, dx)
        a = (at / np.pi + 1.0) * 0.5                                       # float32
        rgb = np.zeros(a.shape + (3,), np.float64)
        rgb[..., 0] = a * 6.0
        rgb[..., 1] = a * 6.0 + 4.0
        rgb[..., 2] = a * 6.0 + 2.0
        rgb = np.clip(np.abs(np.mod(rgb, 6.0) - 3.0) - 1.0, 0.0, 1.0)
        for c in range(3):
            rgb[..., c] = rgb[..., c] * rad + (1.0 - rad)
        out = (rgb * 255).astype(np.uint8)
    return out, mx


def remap_linear_const(img: np.ndarray, map_xy: np.ndarray) -> np.ndarray:
    """cv2.remap(img, map_xy, None, INTER_LINEAR, borderMode=BORDER_CONSTANT) for float32 img [h, w, c] and a
    float32 absolute-coordinate map [h, w, 2] (bands/common/flow.py:19-26 warp_flow).  PARITY UNPINNED: OpenCV is
    absent; this restates its published algorithm - coordinates quantised to 1/32 pixel by a round-half-even
    `cvRound(x * 32)`, integer part by arithmetic shift, the 2x2 weights are products of the float32 tables
    (1 - k/32, k/32), taps outside the image contribute the border value 0, accumulation in float32."""
    h, w = img.shape[:2]
    q = np.rint(map_xy.astype(np.float32) * np.float32(32.0)).astype(np.int64)
    ix, iy = q[..., 0] >> 5, q[..., 1] >> 5
    fx = (q[..., 0] & 31).astype(np.float32) * np.float32(1.0 / 32)
    fy = (q[..., 1] & 31).astype(np.float32) * np.float32(1.0 / 32)
    tx = (np.float32(1.0) - fx, fx)
    ty = (np.float32(1.0) - fy, fy)
    out = None
    for k1 in range(2):
        for k2 in range(2):
            yy, xx = iy + k1, ix + k2
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            v = np.where(ok[..., None], img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], np.float32(0.0))
            term = (v * (ty[k1] * tx[k2])[..., None]).astype(np.float32)
            out = term if out is None else (out + term).astype(np.float32)
    return out


def compute_fwdbwd_mask(fwd: np.ndarray, bwd: np.ndarray, alpha_1: float = 0.05, alpha_2: float = 0.5):
    """bands/common/flow.py:28-40: warp the opposite flow by this one, keep pixels whose round trip closes to
    within alpha_1 (|f| + |f'|) + alpha_2.  float32 throughout, as numpy evaluates the reference."""
    def norm(a):
        return np.sqrt((a[..., 0] * a[..., 0] + a[..., 1] * a[..., 1]).astype(np.float32)).astype(np.float32)

    def one(f, other):
        h, w = f.shape[:2]
        grid = f.astype(np.float32).copy()
        grid[..., 0] += np.arange(w)
        grid[..., 1] += np.arange(h)[:, None]
        warped = remap_linear_const(other.astype(np.float32), grid)
        err = norm(f + warped)
        return err < np.float32(alpha_1) * (norm(f) + norm(warped)) + np.float32(alpha_2)

    return one(fwd, bwd), one(bwd, fwd)
