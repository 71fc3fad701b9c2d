"""Per-launch algorithmic FLOPs / bytes of the bench's GEMM-shaped launches, from the layer shapes alone,
so that bench.py's `roofline.flop_per_launch` and per-family totals can be re-derived by hand.  A bench family is
`<band>/<kernel symbol as rocprofv3 prints it>`; the symbol of a layer follows from launch_gemm's rule (csrc/gemm.hip): the 256 x 256
ping-pong kernel when N % 256 == 0 (or, round 4, >= 192) and there are >= 256 tiles, the 256 x 64 tile for N <= 64 (round 3: the halo-tiled direct kernel for the
3x3 64 -> 64 layers on mx3 maps), else the 128 x 128 tile; the last
template parameter says whether the launch carries MX-fp8 residual tiles (split mode; always false in the f16 mode).
FLOPs = 2 M N K of the layer (real channels: padding and the residual passes not counted); bytes = A once + W once + output once.
python tools/flop_table.py [batch] [H W]"""
import sys

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 and sys.argv[2].isdigit() else (1080, 1920)
rows = []
DENSE, CONV = 0, 1
STD, RESID, QKV, PIXSHUF, PATCH, HEAD, F32 = range(7)


def symbol(amode, epi, M, N, mx, halo=False):
    """launch_gemm(TILE_AUTO) + the launchers' names (csrc/gemm.hip, gemm_kernels.h); BUFP (LDS-DMA through the buffer path) is true
    unless an operand exceeds the 4 GB a buffer resource can address (the N = 32 head conv at batch 32)."""
    m = "true" if mx else "false"
    if epi == HEAD:
        return f"gemm_kernel<256, 32, 4, 1, 1, 5, false, 2, {m}>"
    if epi == PATCH:                                     # engine.hip launches the patch embed on the 128 x 128 tile
        return f"gemm_kernel<128, 128, 2, 2, 0, 4, true, 2, {m}>"
    if N % 256 == 0 and (M // 256) * (N // 256) >= 256:
        return f"gemm8_kernel<{amode}, {epi}, 0, true, {m}>"
    if epi == STD and N % 256 >= 192 and (M // 256) * ((N + 255) // 256) >= 256:      # round 4: N = 192 (RAFT convc2) on the wide tile (PB_TILE_WIDE)
        return f"gemm8_kernel<{amode}, {epi}, 0, true, {m}>"
    if epi == STD and N <= 64:
        if halo:                                         # 3x3, 64 -> 64, stride 1 on mx3 maps: the halo-tiled direct kernel (halo_conv.hip)
            return "conv3x3_c64_mx2_kernel"
        return f"gemm_kernel<256, 64, 4, 1, {amode}, {epi}, true, 2, {m}>"
    if amode == CONV and epi == STD and 64 < N <= 96:     # round 6: RAFT / GMFlow encoder stage 2 on the 128 x 96 tile (PB_TILE_N96)
        return f"gemm_kernel<128, 96, 4, 1, {amode}, {epi}, true, 2, {m}>"
    return f"gemm_kernel<128, 128, 2, 2, {amode}, {epi}, true, 2, {m}>"


def add(band, kind, name, M, N, K, n=1, in_b=2, out_b=2, mx=True, n_launch=None, per_call=False):
    # n_launch: the N the launch is made with when it differs from the layer's (padded volume rows)
    halo = not isinstance(kind, str) and kind == (CONV, STD) and mx and N == 64 and K == 9 * 64 and " s2" not in name
    fam = kind if isinstance(kind, str) else symbol(kind[0], kind[1], M, n_launch or N, mx, halo)
    # algorithmic bytes as the engines count them (engine_base.hip conv / dense): the input MAP once - not its im2col expansion -, the
    # weights once, the output once
    taps, stride = 1, 1
    if not isinstance(kind, str) and kind[0] == CONV and "space-to-depth" not in name:
        taps = 9 if "3x3" in name else 5 if "1x5" in name else 1
        stride = 2 if " s2" in name else 1
    a_bytes = in_b * M * stride * stride * (K / taps)
    if band == "depth" and not per_call:
        n *= ND                                          # the DPT head of a 32-frame call runs as ND chunks of BD frames (DepthEngine::batch_cap, split mode)
    out_bytes = 4.0 * M if (not isinstance(kind, str) and kind[1] == HEAD) else out_b * M * N      # the fused head writes one float per pixel
    rows.append((band, fam, name, n, M, N, K, 2.0 * M * N * K, a_bytes + 2.0 * N * K + out_bytes))


# ---- depth_anything ViT-L @ 518 x 924 (any 16:9 frame): 37 x 66 patches, 2443 tokens padded to 2448 rows per frame
gh, gw, D, F = 37, 66, 1024, 256
ntp, P = 2448, gh * gw
BD = 16 if B > 16 else B                                 # frames per launch: split-mode maps are capped at 2^31 elements (engine.hip batch_cap)
ND = (B + BD - 1) // BD
B_ALL, B = B, BD                                         # the depth rows below are per launch
# round 3: the ViT runs on the whole call (B_ALL frames) at once - 4.8 rounds of 256 x 256 tiles on its N = 1024 GEMMs instead of 2 x 2.4 -,
# only the DPT head below is chunked (engine.hip run_chunk)
Mv = B_ALL * ntp
add("depth", (DENSE, PATCH), "patch embed (+ pos embed)", B_ALL * P, D, 588, mx=False, per_call=True)
add("depth", (DENSE, QKV), "qkv", Mv, 3 * D, D, 24, per_call=True, mx=False)       # round 3: qkv and fc1 run without a weight residual (PB_VIT_RES)
add("depth", "attention", "softmax(QK^T)V, 16 heads", B_ALL * 16 * 2443, 2443, 64 * 2, 24, per_call=True)        # 4 n h t^2 d
rows[-1] = rows[-1][:8] + (4.0 * B_ALL * 16 * 2443 * 64 * 2,)                                   # q, k, v in + o out, fp16
add("depth", (DENSE, RESID), "proj (+ residual)", Mv, D, D, 24, out_b=8, per_call=True)
add("depth", (DENSE, STD), "fc1 + GELU", Mv, 4 * D, D, 24, per_call=True)
add("depth", (DENSE, RESID), "fc2 (+ residual)", Mv, D, 4 * D, 24, out_b=8, per_call=True)
oc = [256, 512, 1024, 1024]
for i in range(4):
    add("depth", (DENSE, STD), f"projects.{i} 1x1", B * P, oc[i], D)
add("depth", (DENSE, PIXSHUF), "resize_layers.0 convT 4x4", B * P, 16 * oc[0], oc[0])
add("depth", (DENSE, PIXSHUF), "resize_layers.1 convT 2x2", B * P, 4 * oc[1], oc[1])
lh = [4 * gh, 2 * gh, gh, (gh - 1) // 2 + 1]
lw = [4 * gw, 2 * gw, gw, (gw - 1) // 2 + 1]
add("depth", (CONV, STD), "resize_layers.3 3x3 s2", B * lh[3] * lw[3], oc[3], 9 * oc[3])
for i in range(4):
    add("depth", (CONV, STD), f"layer{i + 1}_rn 3x3", B * lh[i] * lw[i], F, 9 * oc[i])
for lv in range(3, -1, -1):
    add("depth", (CONV, STD), f"refinenet{lv + 1} RCU convs 3x3", B * lh[lv] * lw[lv], F, 9 * F, 4 if lv < 3 else 2)
    add("depth", (DENSE, STD), f"refinenet{lv + 1} out_conv 1x1 (before the upsample)", B * lh[lv] * lw[lv], F, F)
add("depth", (CONV, STD), "output_conv1 3x3", B * 4 * lh[0] * lw[0], F // 2, 9 * F)
# round 6: output_conv2's nine tap products as a 1 x 1 GEMM over output_conv1's low-resolution map (288 columns); the bilinear resize to 518 x 924, the tap
# sum, ReLU, the 1 x 1 and the last ReLU are one elementwise pass (engine.h wz_, dpt_tail_kernel).  The reference's formulation (PB_HEAD_TAIL=0) is the
# (CONV, HEAD) launch: B * 518 * 924 rows, N = 32, K = 9 * F / 2.
add("depth", (DENSE, STD), "output_conv2 tap products 1x1 at low resolution (F/2 -> 9 x 32)", B * 4 * lh[0] * lw[0], 288, F // 2)

B = B_ALL
# ---- flow_raft at --scale 0.75: (H, W) -> (sh, sw) padded to /8
sh, sw = round(H * 0.75), round(W * 0.75)
Hp, Wp = (sh + 7) // 8 * 8, (sw + 7) // 8 * 8
Fr, pairs = B, B - 1
h2, w2, h4, w4, h8, w8 = Hp // 2, Wp // 2, Hp // 4, Wp // 4, Hp // 8, Wp // 8
Pp = h8 * w8
for enc in ("fnet", "cnet"):
    Fall = Fr
    # round 6: forward pairs only -> the context network skips the clip's last frame (it is no pair's source frame; raft_engine.hip infer)
    Fr = Fall - 1 if enc == "cnet" else Fall
    add("flow", (CONV, PIXSHUF), f"{enc} stem 7x7 s2 (space-to-depth 3x3, N = 4 x 64)", Fr * h4 * w4, 256, 147)
    add("flow", (CONV, STD), f"{enc} layer1 3x3 64->64 @1/2", Fr * h2 * w2, 64, 9 * 64, 4)
    add("flow", (CONV, STD), f"{enc} layer2.0.conv1 3x3 s2 64->96", Fr * h4 * w4, 96, 9 * 64)
    add("flow", (CONV, STD), f"{enc} layer2 3x3 96->96 @1/4", Fr * h4 * w4, 96, 9 * 96, 3)
    add("flow", (CONV, STD), f"{enc} layer2 downsample 1x1 s2", Fr * h4 * w4, 96, 64)
    add("flow", (CONV, STD), f"{enc} layer3.0.conv1 3x3 s2 96->128", Fr * Pp, 128, 9 * 96)
    add("flow", (CONV, STD), f"{enc} layer3 3x3 128->128 @1/8", Fr * Pp, 128, 9 * 128, 3)
    add("flow", (CONV, STD), f"{enc} layer3 downsample 1x1 s2", Fr * Pp, 128, 96)
    add("flow", (DENSE, STD), f"{enc} conv2 1x1 128->256", Fr * Pp, 256, 128)
    Fr = Fall
def vol_stride(a, b):
    # raft_engine.hip prepare(): targets in 8 x 8 tiles; rounded up to a multiple of 256 when that costs under 2 % (ping-pong kernel)
    p = ((a + 7) // 8 * 8) * ((b + 7) // 8 * 8)
    q = (p + 255) // 256 * 256
    return q if q * 50 <= p * 51 else p


lv = [(h8, w8)]
for _ in range(3):
    lv.append((lv[-1][0] // 2, lv[-1][1] // 2))
for l, (a, b) in enumerate(lv):
    add("flow", "corr_volume_kernel", f"correlation volume level {l} (per pair; volume.hip, row stride {vol_stride(a, b)})", Pp, a * b, 256, pairs)
Mu = pairs * Pp
it = 12
add("flow", (CONV, STD), "convc1 1x1 324->256 (a 1 x 1 conv launch on the channel slice)", Mu, 256, 324, it)
add("flow", (CONV, STD), "convc2 3x3 256->192", Mu, 192, 9 * 256, it)
add("flow", "convf1_kernel", "convf1 7x7 2->128 (direct kernel on the fp32 flow field, round 4)", Mu, 128, 98, it)
add("flow", (CONV, STD), "convf2 3x3 128->64", Mu, 64, 9 * 128, it)
add("flow", (CONV, STD), "motion conv 3x3 256->126", Mu, 126, 9 * 256, it)
# (round 4: the update block's maps carry e4m3 copies and its launches MX residual tiles - PB_MX_UPD, default on; the once-per-call context
# shares below and the fp32-output mask.2 keep two fp16 passes)
# round 3, context hoist (raft_engine.hip load()): the context features' 128 of the GRU's 384 input channels do not change over the
# iterations - their share of every gate is computed once per call, the per-iteration convolutions read [h | motion] (256 channels)
add("flow", (CONV, STD), "GRU z|r context share 1x5 / 5x1 128->256 (once per call)", Mu, 256, 5 * 128, 2, mx=False)
add("flow", (CONV, STD), "GRU q context share 1x5 / 5x1 128->128 (once per call)", Mu, 128, 5 * 128, 2, mx=False)
add("flow", (CONV, STD), "GRU z|r 1x5 / 5x1 [h | motion] 256->256", Mu, 256, 5 * 256, 2 * it)
add("flow", (CONV, STD), "GRU q 1x5 / 5x1 [r h | motion] 256->128", Mu, 128, 5 * 256, 2 * it)
add("flow", (CONV, STD), "flow head conv1 3x3 128->256", Mu, 256, 9 * 128, it)
add("flow", "flow_head2_kernel<true>", "flow head conv2 3x3 256->2 (direct kernel)", Mu, 2, 9 * 256, it)
add("flow", (CONV, STD), "mask.0 3x3 128->256 (last iteration)", Mu, 256, 9 * 128)
add("flow", (DENSE, F32), "mask.2 1x1 256->576", Mu, 576, 256, out_b=4, mx=False)


def totals():
    """{(band, family): [launches per step, FLOPs per step, algorithmic bytes per step]}"""
    tot = {}
    for band, fam, name, n, M, N, K, fl, by in rows:
        t = tot.setdefault((band, fam), [0, 0.0, 0.0])
        t[0] += n; t[1] += n * fl; t[2] += n * by
    return tot


def main():
    print(f"batch {B}, frame {W}x{H}; flow at 0.75: {sw}x{sh} -> network {Wp}x{Hp}, 1/8 grid {w8}x{h8}\n")
    print("| band | kernel symbol (split mode) | layer | launches / step | M | N | K | GFLOP / launch | algorithmic MB / launch |")
    print("|---|---|---|---|---|---|---|---|---|")
    for band, fam, name, n, M, N, K, fl, by in rows:
        print(f"| {band} | {fam} | {name} | {n} | {M} | {N} | {K} | {fl / 1e9:.1f} | {by / 1e6:.0f} |")
    tot = totals()
    print("\n| band / family | launches / step | GFLOP / step | mean GFLOP / launch | mean algorithmic MB / launch |")
    print("|---|---|---|---|---|")
    for (band, fam), (n, fl, by) in sorted(tot.items()):
        print(f"| {band}/{fam} | {n} | {fl / 1e9:.0f} | {fl / n / 1e9:.1f} | {by / n / 1e6:.0f} |")
    d = sum(v[1] for k, v in tot.items() if k[0] == "depth")
    f = sum(v[1] for k, v in tot.items() if k[0] == "flow")
    print(f"\ndepth: {d / 1e9 / B:.1f} GFLOP / frame (SURVEY 8d: 2583.1);  flow: {f / 1e9 / pairs:.1f} GFLOP / pair-direction at {Wp}x{Hp}")


if __name__ == "__main__":
    main()
