"""GPU: every HIP kernel of the band, called through the C ABI (pb_op_*), against the oracle's
arithmetic on the same seeded inputs.  Inputs are pre-rounded to fp16 so that the comparison
isolates the kernel (fp32 accumulate, fp16 store) from input quantisation."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import depth_oracle as O
from prisma_amd import engine, synth

pytestmark = pytest.mark.gpu


def h(x):
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    o = engine.Ops(0)
    yield o
    o.close()


@pytest.mark.parametrize("M,N,K,act,tile", [
    (64, 128, 64, 0, 1), (300, 128, 192, 1, 1), (2448, 384, 1024, 2, 1), (1000, 1024, 640, 0, 1),
    (256, 256, 64, 0, 2), (2448, 3072, 1024, 0, 2), (777, 512, 4096, 2, 2), (513, 64, 128, 0, 1),
    (300, 256, 128, 0, 2), (1000, 512, 192, 1, 2), (515, 768, 576, 0, 2), (4096, 1024, 640, 0, 2),
    (2448, 1024, 1024, 0, 4), (640, 256, 320, 2, 4),
])
def test_gemm(ops, M, N, K, act, tile):
    g = np.random.default_rng(M + N + K)
    A, W, b = h(g.standard_normal((M, K))), h(g.standard_normal((N, K)) / np.sqrt(K)), g.standard_normal(N).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + b
    if act == 1:
        ref = np.maximum(ref, 0)
    elif act == 2:
        ref = F.gelu(torch.from_numpy(ref)).numpy()
    out = ops.gemm(A, W, b, act=act, tile=tile)
    assert relmax(out, ref) < 1.5e-3, relmax(out, ref)


def test_gemm_asymmetric_layout(ops):
    # transpose / row-col swap detector: A = identity block, W asymmetric
    M = N = K = 128
    A = np.eye(M, K, dtype=np.float32)
    W = h(np.arange(N * K, dtype=np.float32).reshape(N, K) % 251 / 64.0)
    out = ops.gemm(A, W, None, tile=1)
    assert relmax(out, W.T) < 1e-3


@pytest.mark.parametrize("rows,D", [(7, 384), (100, 768), (2443, 1024)])
def test_layernorm(ops, rows, D):
    g = np.random.default_rng(rows)
    x = (g.standard_normal((rows, D)) * 2 + 0.5).astype(np.float32)
    gm, bt = (1 + 0.1 * g.standard_normal(D)).astype(np.float32), (0.1 * g.standard_normal(D)).astype(np.float32)
    ref = F.layer_norm(torch.from_numpy(x).double(), (D,), torch.from_numpy(gm).double(), torch.from_numpy(bt).double(), 1e-6).numpy()
    assert relmax(ops.layernorm(x, gm, bt), ref) < 1e-3


@pytest.mark.parametrize("B,Hh,N", [(1, 2, 64), (2, 3, 200), (1, 6, 1370), (1, 2, 2443)])
def test_attention(ops, B, Hh, N):
    g = np.random.default_rng(N)
    q, k, v = [h(g.standard_normal((B, Hh, N, 64)) * s) for s in (2.0, 2.0, 1.0)]
    qd, kd, vd = [torch.from_numpy(t).double() for t in (h(q * 0.125) * 8.0, k, v)]
    a = (qd * 0.125) @ kd.transpose(-1, -2)
    ref = (a.softmax(-1) @ vd).numpy()
    out = ops.attention(q, k, v)
    assert relmax(out, ref) < 3e-3, relmax(out, ref)


def test_attention_spiked_row(ops):
    # forces the online-softmax rescale: one key dominates late in the sequence
    g = np.random.default_rng(5)
    q, k, v = [h(g.standard_normal((1, 1, 300, 64))) for _ in range(3)]
    k[0, 0, 270] = h(q[0, 0, 17] * 6.0)
    a = (torch.from_numpy(q).double() * 0.125) @ torch.from_numpy(k).double().transpose(-1, -2)
    ref = (a.softmax(-1) @ torch.from_numpy(v).double()).numpy()
    assert relmax(ops.attention(q, k, v), ref) < 3e-3


@pytest.mark.parametrize("B,Ci,H,W,Co,ks,stride,relu_in,relu_out", [
    (1, 64, 9, 13, 64, 3, 1, 0, 0), (2, 48, 20, 28, 64, 3, 1, 1, 1), (1, 256, 37, 49, 256, 3, 1, 0, 0),
    (1, 384, 37, 49, 384, 3, 2, 0, 0), (2, 128, 16, 16, 32, 1, 1, 0, 0), (1, 96, 30, 22, 128, 3, 1, 0, 1),
])
def test_conv2d(ops, B, Ci, H, W, Co, ks, stride, relu_in, relu_out):
    g = np.random.default_rng(Ci + H)
    x = h(g.standard_normal((B, Ci, H, W)))
    w = h(g.standard_normal((Co, Ci, ks, ks)) / np.sqrt(Ci * ks * ks))
    b = g.standard_normal(Co).astype(np.float32)
    xin = torch.from_numpy(x).double()
    if relu_in:
        xin = xin.relu()
    ref = F.conv2d(xin, torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=stride, padding=ks // 2)
    if relu_out:
        ref = ref.relu()
    out = ops.conv2d(x, w, b, stride=stride, relu_in=bool(relu_in), relu_out=bool(relu_out))
    assert out.shape == tuple(ref.shape)
    assert relmax(out, ref.numpy()) < 1.5e-3, relmax(out, ref.numpy())


def test_split_k_equals_one_pass(ops):
    """Split-K on the 128 x 128 tile (gemm.h splitk: launches with fewer tiles than CUs park their K slices' accumulators and a second launch adds
    them in slice order + runs the epilogue): same answer as the one-pass launch to fp32 summation-order accuracy, against float64, run to run
    identical, for dense GEMMs (K tiles that do not divide by the factor) and an implicit-GEMM convolution with ragged M / N edges."""
    g = np.random.default_rng(77)
    for (M, N, K, act) in [(2448, 1024, 4096, 0), (300, 256, 1088, 1), (515, 384, 2304, 2), (2448, 1024, 1024, 0)]:
        A, W, b = h(g.standard_normal((M, K))), h(g.standard_normal((N, K)) / np.sqrt(K)), g.standard_normal(N).astype(np.float32)
        ref = A.astype(np.float64) @ W.astype(np.float64).T + b
        if act == 1:
            ref = np.maximum(ref, 0)
        elif act == 2:
            ref = F.gelu(torch.from_numpy(ref)).numpy()
        one = ops.gemm(A, W, b, act=act, tile=1)
        ops.set_option("op_splitk", 1)
        try:
            two, again = ops.gemm(A, W, b, act=act, tile=1), ops.gemm(A, W, b, act=act, tile=1)
        finally:
            ops.set_option("op_splitk", 0)
        assert np.array_equal(two, again)
        assert relmax(two, ref) < 1.5e-3 and relmax(two, one) < 1e-3, (M, N, K, relmax(two, ref), relmax(two, one))
    ops.set_option("conv_tile", 1)
    try:
        for (B, Ci, H, W, Co, stride) in [(1, 1024, 19, 33, 256, 1), (1, 512, 37, 66, 1024, 2), (1, 256, 21, 30, 72, 1)]:
            x = h(g.standard_normal((B, Ci, H, W)))
            w = h(g.standard_normal((Co, Ci, 3, 3)) / np.sqrt(Ci * 9))
            b = g.standard_normal(Co).astype(np.float32)
            ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=stride, padding=1).numpy()
            one = ops.conv2d(x, w, b, stride=stride)
            ops.set_option("op_splitk", 1)
            try:
                two = ops.conv2d(x, w, b, stride=stride)
            finally:
                ops.set_option("op_splitk", 0)
            assert relmax(two, ref) < 1.5e-3 and relmax(two, one) < 1e-3, (Ci, relmax(two, ref), relmax(two, one))
    finally:
        ops.set_option("conv_tile", 0)


def test_conv2d_n96_tile_equals_128_tile(ops):
    """The 128 x 96 tile (gemm.h TILE_128x96: convolutions with 64 < N <= 96 - RAFT / GMFlow encoder stage 2) walks K in the order of the 128 x 128
    tile and its two epilogues (an interleaved pair of column blocks + a single block) do the pair epilogue's arithmetic: same bytes, ragged rows
    and columns included, and within tolerance of torch."""
    g = np.random.default_rng(96)
    for (B, Ci, H, W, Co, ks, stride, relu) in [(2, 64, 40, 56, 96, 3, 2, False), (1, 128, 33, 47, 96, 3, 1, True), (3, 64, 19, 31, 72, 3, 1, False),
                                                (1, 64, 24, 40, 96, 1, 2, False), (1, 192, 17, 23, 88, 3, 1, True)]:
        x = h(g.standard_normal((B, Ci, H, W)))
        w = h(g.standard_normal((Co, Ci, ks, ks)) / np.sqrt(Ci * ks * ks))
        b = g.standard_normal(Co).astype(np.float32)
        ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=stride, padding=ks // 2).numpy()
        if relu:
            ref = np.maximum(ref, 0)
        outs = {}
        for tile in (1, 12):
            ops.set_option("conv_tile", tile)
            try:
                outs[tile] = ops.conv2d(x, w, b, stride=stride, relu_out=relu)
            finally:
                ops.set_option("conv_tile", 0)
        assert np.array_equal(outs[1], outs[12]), (Ci, Co, ks, stride, float(np.abs(outs[1] - outs[12]).max()))
        assert relmax(outs[12], ref) < 1.5e-3, (Ci, Co, relmax(outs[12], ref))


@pytest.mark.parametrize("tile", [2, 4])
def test_conv2d_wide_tiles(ops, tile):
    ops.set_option("conv_tile", tile)
    try:
        g = np.random.default_rng(tile)
        for (B, Ci, H, W, Co, stride) in [(1, 256, 37, 49, 256, 1), (2, 128, 24, 40, 512, 1), (1, 320, 33, 21, 256, 2), (3, 128, 19, 31, 128, 1),
                                          (1, 64, 40, 56, 96, 2)]:
            x = h(g.standard_normal((B, Ci, H, W)))
            w = h(g.standard_normal((Co, Ci, 3, 3)) / np.sqrt(Ci * 9))
            b = g.standard_normal(Co).astype(np.float32)
            ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                           stride=stride, padding=1).numpy()
            out = ops.conv2d(x, w, b, stride=stride)
            assert relmax(out, ref) < 1.5e-3, (tile, Ci, relmax(out, ref))
    finally:
        ops.set_option("conv_tile", 0)


@pytest.mark.parametrize("H,W,OH,OW,align", [(3, 4, 5, 7, 1), (19, 33, 37, 66, 1), (20, 28, 40, 56, 1),
                                              (296, 392, 518, 686, 1), (37, 49, 90, 120, 0)])
def test_bilinear(ops, H, W, OH, OW, align):
    x = h(np.random.default_rng(H).standard_normal((2, 16, H, W)))
    ref = F.interpolate(torch.from_numpy(x), (OH, OW), mode="bilinear", align_corners=bool(align)).numpy()
    assert relmax(ops.bilinear(x, OH, OW, bool(align)), ref) < 1.5e-3


@pytest.mark.parametrize("H,W", [(96, 128), (720, 1280), (1080, 1920), (600, 333)])
def test_preprocess(ops, H, W):
    frame = synth.frames(1, H, W, seed=H)[0]
    ref = O.preprocess(frame)
    out = ops.preprocess(frame)
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() < 2e-4, np.abs(out - ref).max()


def test_encode_depth_bit_exact(ops, golden_dir):
    z = np.load(os.path.join(golden_dir, "encode.npz"))
    rgb, mn, mx = ops.encode_depth(z["pred"], flip=True)
    assert np.array_equal(rgb[0], z["vid"])
    assert mn[0] == z["pred"].min() and mx[0] == z["pred"].max()
    g = np.random.default_rng(3)
    d = (g.standard_normal((3, 270, 480)) * 4).astype(np.float32)
    rgb, mn, mx = ops.encode_depth(d, flip=False)
    for i in range(3):
        ref, lo, hi = O.encode_depth_video(d[i], flip=False)
        assert np.array_equal(rgb[i], ref) and mn[i] == lo and mx[i] == hi
    # degenerate frame: max == min -> NaN -> 0 like the reference
    rgb, mn, mx = ops.encode_depth(np.full((1, 8, 8), 2.5, np.float32))
    assert not rgb.any() and mn[0] == mx[0] == 2.5


@pytest.mark.parametrize("B,L,masked", [(2, 80, False), (3, 391, True), (1, 4590, True)])
def test_attention128_single_head(ops, B, L, masked):
    """attention128.hip against float64 torch: softmax(q k^T / sqrt(128) + mask) v, one head of 128 (bands/gmflow/transformer.py:8-15,
    47-101); L = 80 is one 8 x 10 window of the small golden, 391 an odd length with a key tail, 4590 the 51 x 90 window of 1080p x 0.75.
    The mask is the shifted-window one (:18-44): -100 on the logit of a key from another region."""
    import torch
    g = np.random.default_rng(L)
    q = (g.standard_normal((B, L, 128)) * 1.5).astype(np.float32)
    k = (g.standard_normal((B, L, 128)) * 1.5).astype(np.float32)
    v = g.standard_normal((B, L, 128)).astype(np.float32)
    region = (g.integers(0, 3, (B, L)).astype(np.int8) if masked else None)
    got = ops.attention128(q, k, v, region)
    tq, tk, tv = (torch.from_numpy(t.astype(np.float16).astype(np.float64)) for t in (q, k, v))
    s = tq @ tk.transpose(1, 2) / 128 ** 0.5
    if masked:
        r = torch.from_numpy(region.astype(np.int64))
        s = s + torch.where(r[:, :, None] != r[:, None, :], -100.0, 0.0)
    ref = (torch.softmax(s, -1) @ tv).numpy()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print("\n  attention128 B %d L %d masked %s: max err / range %.2e" % (B, L, masked, err))
    assert err < 1e-3                                      # P is rounded to fp16 before P V: ~2e-4


@pytest.mark.parametrize("B,L,vc,masked,kxor", [(4, 80, 128, True, 0), (2, 391, 128, False, 1), (2, 1064, 32, False, 0), (8, 266, 128, True, 4), (1, 4590, 32, False, 0),
                                                (4, 4590, 128, True, 0), (1, 8300, 128, True, 0)])      # 8300 > 8192: the region ids no longer fit the LDS (global-memory path)
def test_attention128_split_precision(ops, B, L, vc, masked, kxor):
    """attention128.hip in the flow_gmflow band's split precision against float64 torch on UNROUNDED operands: hi + lo fp16 pairs for q, k, v
    and the probabilities leave ~2^-21 relative operand error, so the result sits at fp32 accumulation noise (the single-pass kernel above
    is compared with fp16-rounded operands and still shows 2e-4).  vc = 32: V padded to 32 columns (coordinates / flow); kxor: keys and
    values from the partner batch element (cross attention); region [4, L]: window masks shared by the batch (b % 4)."""
    import torch
    g = np.random.default_rng(L + vc)
    q = (g.standard_normal((B, L, 128)) * 1.5).astype(np.float32)
    k = (g.standard_normal((B, L, 128)) * 1.5).astype(np.float32)
    v = (g.standard_normal((B, L, vc)) * (40.0 if vc == 32 else 1.0)).astype(np.float32)
    nreg = 4 if masked else 0
    region = g.integers(0, 3, (nreg, L)).astype(np.int8) if masked else None
    got = ops.attention128_split(q, k, v, region, kxor)
    tq, tk, tv = (torch.from_numpy(t.astype(np.float64)) for t in (q, k, v))
    idx = torch.arange(B) ^ kxor
    s = tq @ tk[idx].transpose(1, 2) / 128 ** 0.5
    if masked:
        r = torch.from_numpy(region.astype(np.int64))[torch.arange(B) % 4]
        s = s + torch.where(r[:, :, None] != r[:, None, :], -100.0, 0.0)
    ref = (torch.softmax(s, -1) @ tv[idx]).numpy()
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print("\n  attention128 split B %d L %d vcols %d masked %s kxor %d: max err / range %.2e" % (B, L, vc, masked, kxor, err))
    assert err < 2e-5


def test_encode_still_bytes_match_the_reference(ops, golden_dir):
    """SURVEY 8 a-1.10 / f-3 on the GPU: pb_depth_encode_still against what the REAL reference write_depth (bands/common/io.py:138-172,
    encode.py:73-95,141-146) handed to cv2.imwrite for the same depth map (tests/golden/write_depth.npz, made by oracle/make_golden.py
    write_depth): every byte - heat ramp, Sobel-edge saturation, min / max packed in pixels (0,0), (0,1), uint8 truncation - relative
    (flipped) and metric (not flipped); and against the band's own host restatement on odd shapes (1-pixel borders, ragged sizes)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "bands"))
    from common import io as IO
    import depth_anything as band
    from PIL import Image
    z = np.load(os.path.join(golden_dir, "write_depth.npz"))
    depth = np.ascontiguousarray(z["depth"], np.float32)
    for name, flip in (("rel_rgb", True), ("met_rgb", False)):
        rgb, lo, hi = ops.encode_still(depth, flip=flip)
        assert rgb.shape == z[name].shape
        assert np.array_equal(rgb, z[name]), (name, int((rgb != z[name]).sum()))
        assert lo == float(depth.min()) and hi == float(depth.max())
    dec = lambda px: (float(px[0]) + float(px[1]) * 256 + float(px[2]) * 65536) / (256 ** 3 - 1) * 1000.0   # viewer contract, view.py:186-210
    assert abs(dec(rgb[0, 0]) - depth.min()) < 1e-4 and abs(dec(rgb[0, 1]) - depth.max()) < 1e-4
    g = np.random.default_rng(9)
    for (H, W) in ((1, 7), (5, 2), (33, 47), (90, 160), (1080, 1920)):
        d = (g.standard_normal((H, W)).cumsum(1).cumsum(0) * 0.37 + 5.0).astype(np.float32)
        for flip in (True, False):
            rgb, _, _ = ops.encode_still(d, flip=flip)
            import tempfile
            with tempfile.TemporaryDirectory() as t:
                p = os.path.join(t, "a.png")
                IO.write_depth(p, d.copy(), band.heat_to_rgb, normalize=True, flip=flip, heatmap=True, encode_range=True)
                want = np.asarray(Image.open(p))
            assert np.array_equal(rgb, want), (H, W, flip, int((rgb != want).sum()))


def test_epilogue_report_names_launch_kinds_outside_the_fast_list():
    """PB_EPI_REPORT=1 (gemm.hip): an EPI_STD launch whose activation / skip combination has no straight-line epilogue copy is named once on
    stderr; the kinds the bands launch are all inside the list (profiles/r05u_epilogue_report.txt: nothing reported by the depth, flow_raft,
    flow_gmflow, mask and metric-depth suites)."""
    import subprocess, sys
    code = ("import numpy as np; from prisma_amd import engine; o = engine.Ops(0); A = np.ones((256, 64), np.float32); W = np.ones((128, 64), np.float32);"
            "o.gemm(A, W, None, act=1, tile=2); o.gemm(A, W, None, act=3, tile=2); o.gemm(A, W, None, act=3, tile=2); o.close()")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PB_EPI_REPORT="1"),
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-400:]
    lines = [l for l in r.stderr.splitlines() if "pb_epi_report" in l]
    assert len(lines) == 1 and "key 3 " in lines[0] and "act 3" in lines[0], r.stderr[-400:]


@pytest.mark.parametrize("M,N,ldo", [(391, 392, 576), (128, 64, 64), (77, 136, 192), (300, 1288, 1344)])
def test_corr_volume_kernel_values_and_untouched_memory(M, N, ldo):
    """volume.hip on its own (pb_op_corr_volume): every entry of the [M, N] volume against fp32 numpy on the fp16-rounded operands, and
    nothing written outside it - the padding columns of a row and 32 guard rows behind the last one stay NaN.  The kernel's stores carry
    their row in the scalar offset of a buffer store (documented as outside the bounds check): last row tiles with fewer than 32 valid
    rows (M = 391, 77, 300) mask them in the lane offset."""
    rng = np.random.default_rng(M * 7 + N)
    A = rng.standard_normal((M, 256)).astype(np.float32) * 0.25
    W = rng.standard_normal((N, 256)).astype(np.float32) * 0.25
    o = engine.Ops(0)
    out = o.corr_volume(A, W, ldo=ldo, guard_rows=32)
    o.close()
    ref = A.astype(np.float16).astype(np.float32) @ W.astype(np.float16).astype(np.float32).T
    got = out[:M, :N]
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-3          # fp16 output rounding
    assert np.isnan(out[:M, N:]).all(), "padding columns written"
    assert np.isnan(out[M:]).all(), "rows past M written"
