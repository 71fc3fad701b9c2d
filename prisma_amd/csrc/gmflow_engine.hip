// GmflowEngine: the flow_gmflow band on one MI355X.
// Reference call stack being replaced: bands/flow_gmflow.py:134-157 (frame loop: cv2.resize(fx = fy = scale, INTER_CUBIC)) -> :66-118
// infer (InputPadder(padding_factor = 16), GMFlow(...), unpad) -> bands/gmflow/gmflow.py:95-170 -> backbone.py:56-117 (CNNEncoder),
// utils.py:53-86 (normalise, per-window sine positions), transformer.py:108-290 (6 x [self-attention, cross-attention + FFN] over
// 2 x 2 shifted windows), matching.py:7-42 (global correlation softmax), transformer.py:300-337 (flow propagation), gmflow.py:74-92
// (convex upsampling) -> bands/common/flow.py:64-88 + encode.py:98-126 (process_flow).
//
// Schedule, not arithmetic:
//   * every frame passes the backbone once per sequence (the reference encodes both frames of every pair);
//   * the token stream [pairs, 2 frames, P, 128] is fp32; what enters a block's cross attention as `target` is the OTHER frame's stream
//     as it entered the block (transformer.py:286-288), so all five projections that read the entering stream (q, k, v of the self
//     attention; k, v of the cross attention) are ONE GEMM with N = 640, and the cross attention reads the partner frame's windows
//     through the attention kernel's batch-index xor (attention128.hip kxor) - nothing is concatenated or swapped in memory;
//   * window split / roll / merge (transformer.py:47-101) are index maps inside the pack and LayerNorm kernels (gmflow_kernels.hip);
//   * global matching and propagation are the same flash-style attention with V = pixel coordinates / flow padded to 32 columns: the
//     P x P correlation volume (2.7 GB fp32 per pair at 1080p x 0.75) never exists;
//   * PB_PREC_SPLIT: every MFMA operand is a hi + lo fp16 pair (GEMMs: three K segments; attention: three passes for S and for P V) -
//     this network's two softmax stages amplify operand rounding several times more than RAFT does.
#include "gmflow_engine.h"

#include <math.h>
#include <string.h>

#include <algorithm>

int GmflowEngine::upload(const std::string &name, int n, float **dst) {
    const pb_tensor *t = find(name);
    PB_CHECK(t && t->shape[0] == n, PB_ERR_ARG, "missing weight '%s' [%d]", name.c_str(), n);
    void *p = nullptr;
    PB_HIP(hipMalloc(&p, (size_t)n * 4));
    owned_.push_back(p);
    PB_HIP(hipMemcpy(p, t->data, (size_t)n * 4, hipMemcpyHostToDevice));
    *dst = (float *)p;
    return 0;
}

int GmflowEngine::load(const pb_tensor *w, int n) {
    int r = begin_load(w, n);
    if (r) return r;
    mx_ = 0;                                    // fp16 residual parts everywhere: activations AND weights split, three K segments
    pack_tapin_ = 0;
    if ((r = pack_encoder("backbone", false, false, fnet_))) return r;
    if ((r = pack_conv("backbone.conv2", true, nullptr, nullptr, fnet_.out, 1))) return r;
    auto lin = [&](const std::string &name, int N, int K, PackedW &out, bool bias) -> int {
        const pb_tensor *t = find(name + ".weight");
        PB_CHECK(t && t->ndim == 2 && t->shape[0] == N && t->shape[1] == K, PB_ERR_ARG, "missing linear '%s' [%d, %d]", name.c_str(), N, K);
        const float *b = nullptr;
        if (bias) {
            const pb_tensor *tb = find(name + ".bias");
            PB_CHECK(tb && tb->shape[0] == N, PB_ERR_ARG, "missing bias of '%s'", name.c_str());
            b = (const float *)tb->data;
        }
        return pack((const float *)t->data, N, K, K, out, b, 1, 1);
    };
    for (int i = 0; i < 6; ++i) {
        Layer &L = layers_[i];
        const std::string p = "transformer.layers." + std::to_string(i) + ".", s = p + "self_attn.", c = p + "cross_attn_ffn.";
        std::vector<float> w1((size_t)640 * 128);
        const char *parts[5] = {"q_proj", "k_proj", "v_proj", "k_proj", "v_proj"};
        for (int j = 0; j < 5; ++j) {
            const std::string nm = (j < 3 ? s : c) + parts[j] + ".weight";
            const pb_tensor *t = find(nm);
            PB_CHECK(t && t->ndim == 2 && t->shape[0] == 128 && t->shape[1] == 128, PB_ERR_ARG, "missing linear '%s' [128, 128]", nm.c_str());
            memcpy(w1.data() + (size_t)j * 128 * 128, t->data, (size_t)128 * 128 * 4);
        }
        if ((r = pack(w1.data(), 640, 128, 128, L.w1, nullptr, 1, 1))) return r;
        if ((r = lin(s + "merge", 128, 128, L.merge_s, false))) return r;
        if ((r = lin(c + "q_proj", 128, 128, L.q_c, false))) return r;
        if ((r = lin(c + "merge", 128, 128, L.merge_c, false))) return r;
        if ((r = lin(c + "mlp.0", 1024, 256, L.mlp0, false))) return r;
        if ((r = lin(c + "mlp.2", 128, 1024, L.mlp2, false))) return r;
        if ((r = upload(s + "norm1.weight", 128, &L.ln1s_g)) || (r = upload(s + "norm1.bias", 128, &L.ln1s_b))) return r;
        if ((r = upload(c + "norm1.weight", 128, &L.ln1c_g)) || (r = upload(c + "norm1.bias", 128, &L.ln1c_b))) return r;
        if ((r = upload(c + "norm2.weight", 128, &L.ln2c_g)) || (r = upload(c + "norm2.bias", 128, &L.ln2c_b))) return r;
    }
    if ((r = lin("feature_flow_attn.q_proj", 128, 128, ffq_, true))) return r;
    if ((r = lin("feature_flow_attn.k_proj", 128, 128, ffk_, true))) return r;
    if ((r = pack_conv("upsampler.0", true, nullptr, nullptr, up0_, 1))) return r;
    if ((r = pack_conv("upsampler.2", true, nullptr, nullptr, up2_, 1))) return r;
    tmap_.clear();
    PB_HIP(hipDeviceSynchronize());
    return 0;
}

namespace {
// PositionEmbeddingSine(num_pos_feats = 64, temperature 10000, normalize, scale 2 pi) of ONE wh x ww window (position.py:26-46), tiled over
// the 2 x 2 windows (utils.py:61-86), as a token-major table [h8 * w8, 128]: channels 0..63 from y, 64..127 from x, (sin, cos) interleaved
void sine_positions(int h8, int w8, std::vector<float> &pos) {
    const int wh = h8 / 2, ww = w8 / 2;
    const float eps = 1e-6f, scale = 6.283185307179586f;
    pos.assign((size_t)h8 * w8 * 128, 0.f);
    float dim_t[64];
    for (int i = 0; i < 64; ++i) dim_t[i] = powf(10000.f, 2.f * (float)(i / 2) / 64.f);
    for (int y = 0; y < h8; ++y)
        for (int x = 0; x < w8; ++x) {
            const float ye = (float)(y % wh + 1) / ((float)wh + eps) * scale, xe = (float)(x % ww + 1) / ((float)ww + eps) * scale;
            float *p = pos.data() + ((size_t)y * w8 + x) * 128;
            for (int i = 0; i < 64; ++i) {
                const float ay = ye / dim_t[i], ax = xe / dim_t[i];
                p[i] = (i & 1) ? cosf(ay) : sinf(ay);
                p[64 + i] = (i & 1) ? cosf(ax) : sinf(ax);
            }
        }
}

// region ids of generate_shift_window_attn_mask (transformer.py:18-44) in window order: [4 windows][Lw]
void shift_regions(int h8, int w8, std::vector<int8_t> &reg) {
    const int wh = h8 / 2, ww = w8 / 2;
    reg.assign((size_t)4 * wh * ww, 0);
    for (int win = 0; win < 4; ++win)
        for (int ly = 0; ly < wh; ++ly)
            for (int lx = 0; lx < ww; ++lx) {
                const int ry = (win >> 1) * wh + ly, rx = (win & 1) * ww + lx;
                const int cy = ry < h8 - wh ? 0 : (ry < h8 - wh / 2 ? 1 : 2), cx = rx < w8 - ww ? 0 : (rx < w8 - ww / 2 ? 1 : 2);
                reg[((size_t)win * wh + ly) * ww + lx] = (int8_t)(cy * 3 + cx);
            }
}
}  // namespace

int GmflowEngine::prepare_g(int F, int H, int W, float scale, int dirs) {
    if (F <= gF_ && H == gH_ && W == gW_ && scale == gS_ && dirs <= gD_) return 0;
    PB_HIP(hipStreamSynchronize(stream));
    gF_ = 0; gH_ = 0; gW_ = 0; gD_ = 0;       // (a failure below must not leave the old plan's key on a half-written geometry: raft_engine.hip prepare)
    geometry(H, W, scale, 16);
    if (isz_h_ > 0) {                 // --inference_size: the network's size is given, nothing is padded (sh_, sw_ stay the output size)
        padl_ = padt_ = 0;
        Hp_ = isz_h_; Wp_ = isz_w_;
        h8_ = Hp_ / 8; w8_ = Wp_ / 8; P_ = h8_ * w8_;
    }
    PB_CHECK(h8_ >= 4 && w8_ >= 4 && h8_ % 2 == 0 && w8_ % 2 == 0, PB_ERR_ARG, "flow_gmflow: %dx%d is too small", sh_, sw_);
    g_.h8 = h8_; g_.w8 = w8_; g_.P = P_; g_.wh = h8_ / 2; g_.ww = w8_ / 2; g_.Lw = g_.wh * g_.ww; g_.ldv = (int)round_up(g_.Lw, 32);
    ldvP_ = (int)round_up(P_, 32);
    const int NP = F - 1;
    const int64_t R = (int64_t)NP * 2 * P_, B = (int64_t)NP * dirs, Bw = (int64_t)NP * 8;
    const size_t slack = 1 << 16;
    for (int pass = 0; pass < 2; ++pass) {
        planning_ = pass == 0;
        arena_off_ = 0;
        carve_encoder(F);
        feat_ = (float *)carve((size_t)F * P_ * 128 * 4);
        pos_ = (float *)carve((size_t)P_ * 128 * 4);
        region_ = (int8_t *)carve((size_t)4 * g_.Lw);
        X_ = (float *)carve((size_t)R * 128 * 4); blk0_ = (float *)carve((size_t)R * 128 * 4);
        Xs_ = (f16 *)carve((size_t)R * 256 * 2 + slack);
        Y1_ = (float *)carve((size_t)R * 640 * 4); Yq_ = (float *)carve((size_t)R * 128 * 4);
        Qw_ = (f16 *)carve((size_t)R * 256 * 2 + slack); Kw_ = (f16 *)carve((size_t)R * 256 * 2 + slack); Kcw_ = (f16 *)carve((size_t)R * 256 * 2 + slack);
        Vtw_ = (f16 *)carve((size_t)Bw * 2 * 128 * g_.ldv * 2 + slack); Vtcw_ = (f16 *)carve((size_t)Bw * 2 * 128 * g_.ldv * 2 + slack);
        Ow_ = (float *)carve((size_t)R * 128 * 4); Os_ = (f16 *)carve((size_t)R * 256 * 2 + slack);
        M_ = (float *)carve((size_t)R * 128 * 4);
        cat_ = (f16 *)carve((size_t)R * 512 * 2 + slack); Hs_ = (f16 *)carve((size_t)R * 2048 * 2 + slack);
        gridvt_ = (f16 *)carve((size_t)64 * ldvP_ * 2 + slack);
        Om_ = (float *)carve((size_t)B * P_ * 32 * 4);
        flowm_ = (float *)carve((size_t)B * P_ * 2 * 4); flowp_ = (float *)carve((size_t)B * P_ * 2 * 4);
        Vtf_ = (f16 *)carve((size_t)B * 64 * ldvP_ * 2 + slack);
        qs_ = (f16 *)carve((size_t)R * 256 * 2 + slack); ks_ = (f16 *)carve((size_t)R * 256 * 2 + slack);
        umap_ = (f16 *)carve((size_t)B * P_ * 384 * 2 + slack); u1_ = (f16 *)carve((size_t)round_up(B * P_, 256) * 512 * 2 + slack);
        gmask_ = (float *)carve((size_t)B * P_ * 576 * 4);
        gup_ = (float *)carve((size_t)B * sh_ * sw_ * 2 * 4);
        gupi_ = isz_h_ > 0 ? (float *)carve((size_t)B * Hp_ * Wp_ * 2 * 4) : nullptr;
        gmaxd_ = (unsigned *)carve((size_t)B * 4);
        if (pass == 0) {
            const int rc = commit_arena("flow_gmflow");
            if (rc) return rc;
        }
    }
    std::vector<float> pos;
    std::vector<int8_t> reg;
    sine_positions(h8_, w8_, pos);
    shift_regions(h8_, w8_, reg);
    PB_HIP(hipMemcpyAsync(pos_, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, stream));
    PB_HIP(hipMemcpyAsync(region_, reg.data(), reg.size(), hipMemcpyHostToDevice, stream));
    int r = launch_gm_grid_vt(stream, gridvt_, P_, w8_, ldvP_);
    if (r) return r;
    if ((r = upload_resize_tables(H, W, scale))) return r;         // synchronises: the host vectors above stay alive until then
    gF_ = F; gH_ = H; gW_ = W; gS_ = scale; gD_ = dirs;
    return 0;
}

int GmflowEngine::gemm32(const f16 *A, int lda, int64_t M, const PackedW &w, float *out, int ldo) {
    GemmArgs a;
    a.A = A; a.lda = lda; a.N = w.N; a.M = (int)M;
    set_weights(a, w, false);
    a.out32 = out; a.ldo = ldo; a.scale = 1.f;
    tic(F_GEMM, 2.0 * M * (double)w.N * w.Kreal, 2.0 * ((double)M * w.Kreal + (double)w.N * w.Kreal) + 4.0 * M * w.N, 1.0 + w.sa + w.sw);
    int r = launch_gemm(cur_, A_DENSE, EPI_F32, TILE_AUTO, a);
    if (timer.enabled && !r) timer.recs[open_.back()].name = pb_gemm_last_kernel();
    toc();
    return r;
}

int GmflowEngine::gemm16(const f16 *A, int lda, int64_t M, const PackedW &w, f16 *out, int ldo, int act, int lo_off) {
    GemmArgs a;
    a.A = A; a.lda = lda; a.N = w.N; a.M = (int)M;
    set_weights(a, w, false);
    a.out = out; a.ldo = ldo; a.act = act; a.lo_off = lo_off;
    tic(F_GEMM, 2.0 * M * (double)w.N * w.Kreal, 2.0 * ((double)M * w.Kreal + (double)w.N * w.Kreal + (double)M * w.N), 1.0 + w.sa + w.sw);
    int r = launch_gemm(cur_, A_DENSE, EPI_STD, TILE_AUTO, a);
    if (timer.enabled && !r) timer.recs[open_.back()].name = pb_gemm_last_kernel();
    toc();
    return r;
}

int GmflowEngine::attention(const Attn128Args &a, double keys_per_query) {
    // S and P V: 2 x 2 x (128 + vcols) flops per (query, key); three MFMA passes each in split mode
    const double qk = 2.0 * a.B * (double)a.L * keys_per_query * 128.0, pv = 2.0 * a.B * (double)a.L * keys_per_query * a.vcols;
    tic(F_ATTN, qk + pv, 0, ((a.split ? 3.0 : 1.0) * qk + (a.split && !a.pv_single ? 3.0 : 1.0) * pv) / (qk + pv));
    int r = launch_attention128x(cur_, a);
    toc();
    return r;
}

int GmflowEngine::set_inference_size(int h, int w) {
    PB_CHECK((h == 0 && w == 0) || (h >= 32 && w >= 32 && h % 16 == 0 && w % 16 == 0), PB_ERR_ARG,
             "flow_gmflow: --inference_size %d %d must be multiples of 16 (8 x the 2 x 2 window split; the reference fails in its window split otherwise)", h, w);
    if (h != isz_h_ || w != isz_w_) { isz_h_ = h; isz_w_ = w; gF_ = 0; }      // re-plan on the next call
    return 0;
}

int GmflowEngine::infer(const uint8_t *frames, int F, int H, int W, float scale, int /*iters*/, int backward, float *flow_out,
                        uint8_t *rgb_out, float *maxdisp, uint8_t *mask_out, float alpha1, float alpha2) {
    PB_CHECK(frames && F >= 2 && H > 0 && W > 0 && scale > 0.f, PB_ERR_ARG, "flow_gmflow infer: bad arguments");
    PB_CHECK(!mask_out || backward, PB_ERR_ARG, "consistency masks need both directions (backward = 1)");
    PB_HIP(hipSetDevice(device));
    const int dirs = backward ? 2 : 1;
    int r = prepare_g(F, H, W, scale, dirs);
    if (r) return r;
    timer.reset();
    fstages_.clear();
    const int NP = F - 1, B = NP * dirs, Bw = NP * 8, P = P_;
    const int64_t R = (int64_t)NP * 2 * P;
    const int split = split_w_ ? 1 : 0;
    const int es = split_w_ ? 2 : 1;
    last_nd_ = B;

    // ---- frame prep (resize, replicate pad to /16, ImageNet normalisation) and the backbone, once per frame ----
    tic(F_PP, 0, (double)F * H * W * 3);
    r = launch_raft_prep(stream, frames, F, H, W, sh_, sw_, Hp_, Wp_, padl_, padt_, scale != 1.f, xi_, xc_, yi_, yc_, img_, nullptr, 1,
                         split_w_ ? 64 : 0, -1, 1, isz_h_ > 0);
    toc();
    if (r) return r;
    const f16 *x = nullptr;
    if ((r = run_encoder(fnet_, true, F, &x))) return r;
    if ((r = gemm32(x, es * 128, (int64_t)F * P, fnet_.out, feat_, 128))) return r;
    fstages_["feat"] = FStage{feat_, F, P, 128};

    // ---- tokens + per-window sine positions; the six transformer blocks ----
    tic(F_ELT, 0, 0);
    r = launch_gm_tokens(stream, feat_, pos_, X_, Xs_, NP, P);
    toc();
    if (r) return r;
    for (int li = 0; li < 6; ++li) {
        const Layer &L = layers_[li];
        const int shifted = li & 1;
        if ((r = gemm32(Xs_, 256, R, L.w1, Y1_, 640))) return r;
        GmPackJobs jobs{};
        jobs.n = 5;
        jobs.j[0] = GmPackJob{Y1_, 640, 0, Qw_, 0};
        jobs.j[1] = GmPackJob{Y1_, 640, 128, Kw_, 0};
        jobs.j[2] = GmPackJob{Y1_, 640, 256, Vtw_, 1};
        jobs.j[3] = GmPackJob{Y1_, 640, 384, Kcw_, 0};
        jobs.j[4] = GmPackJob{Y1_, 640, 512, Vtcw_, 1};
        tic(F_ELT, 0, 0);
        r = launch_gm_pack(stream, jobs, g_, Bw, shifted);
        toc();
        if (r) return r;
        Attn128Args a;
        a.Q = Qw_; a.K = Kw_; a.Vt = Vtw_; a.region = shifted ? region_ : nullptr; a.nreg = 4; a.O = Ow_;
        a.B = Bw; a.L = g_.Lw; a.ldv = g_.ldv; a.split = split; a.vcols = 128; a.ldq = 256; a.v_bstride = (int64_t)2 * 128 * g_.ldv;
        a.pv_single = 1;                  // window attention: P and V as single fp16 (2.6e-4 of the budget), q / k split (attention128.hip)
        if ((r = attention(a, g_.Lw))) return r;
        tic(F_ELT, 0, 0);
        r = launch_gm_split_rows(stream, Ow_, 128, 128, Os_, R);
        toc();
        if (r) return r;
        if ((r = gemm32(Os_, 256, R, L.merge_s, M_, 128))) return r;
        tic(F_LN, 0, 0);
        r = launch_gm_ln(stream, M_, L.ln1s_g, L.ln1s_b, X_, Xs_, R, g_, 1, shifted, 0);
        toc();
        if (r) return r;
        // cross attention + FFN: queries from the updated stream, keys / values from the partner frame's ENTERING stream (Kcw_, Vtcw_)
        if ((r = gemm32(Xs_, 256, R, L.q_c, Yq_, 128))) return r;
        jobs.n = 1;
        jobs.j[0] = GmPackJob{Yq_, 128, 0, Qw_, 0};
        tic(F_ELT, 0, 0);
        r = launch_gm_pack(stream, jobs, g_, Bw, shifted);
        toc();
        if (r) return r;
        a.K = Kcw_; a.Vt = Vtcw_; a.kxor = 4;
        if ((r = attention(a, g_.Lw))) return r;
        tic(F_ELT, 0, 0);
        r = launch_gm_split_rows(stream, Ow_, 128, 128, Os_, R);
        toc();
        if (r) return r;
        if ((r = gemm32(Os_, 256, R, L.merge_c, M_, 128))) return r;
        tic(F_LN, 0, 0);
        r = launch_gm_ln(stream, M_, L.ln1c_g, L.ln1c_b, X_, cat_, R, g_, 1, shifted, 1);
        toc();
        if (r) return r;
        if ((r = gemm16(cat_, 512, R, L.mlp0, Hs_, 2048, ACT_GELU, 1024))) return r;
        if ((r = gemm32(Hs_, 2048, R, L.mlp2, M_, 128))) return r;
        tic(F_LN, 0, 0);
        r = launch_gm_ln(stream, M_, L.ln2c_g, L.ln2c_b, X_, Xs_, R, g_, 0, 0, 0);
        toc();
        if (r) return r;
        if (debug && li == 0) {
            PB_HIP(hipMemcpyAsync(blk0_, X_, (size_t)R * 128 * 4, hipMemcpyDeviceToDevice, stream));
            fstages_["block0"] = FStage{blk0_, (int64_t)NP * 2, P, 128};
        }
    }
    fstages_["tfeat"] = FStage{X_, (int64_t)NP * 2, P, 128};

    // ---- global matching (matching.py:7-42): softmax over ALL target tokens of the dot products, expectation of their coordinates ----
    const int64_t img = (int64_t)P * 256;                    // one frame's rows of a split token matrix
    Attn128Args m;
    m.Q = Xs_; m.O = Om_; m.B = B; m.L = P; m.ldv = ldvP_; m.split = split; m.vcols = 32; m.ldq = 256;
    m.Vt = gridvt_; m.v_shared = 1;
    if (dirs == 2) { m.K = Xs_; m.q_bstride = m.k_bstride = img; m.kxor = 1; }
    else { m.K = Xs_ + img; m.q_bstride = m.k_bstride = 2 * img; }
    if ((r = attention(m, P))) return r;
    tic(F_ELT, 0, 0);
    r = launch_gm_match_flow(stream, Om_, flowm_, Vtf_, B, P, w8_, ldvP_);
    toc();
    if (r) return r;
    fstages_["flow_match"] = FStage{flowm_, B, P, 2};

    // ---- flow propagation (transformer.py:316-337): self-similarity of the source frame's features spreads the matched flow ----
    if ((r = gemm32(Xs_, 256, R, ffq_, Yq_, 128))) return r;
    tic(F_ELT, 0, 0);
    r = launch_gm_split_rows(stream, Yq_, 128, 128, qs_, R);
    toc();
    if (r) return r;
    if ((r = gemm32(qs_, 256, R, ffk_, M_, 128))) return r;         // the key is k_proj of the PROJECTED query, as in the reference (:326-327)
    tic(F_ELT, 0, 0);
    r = launch_gm_split_rows(stream, M_, 128, 128, ks_, R);
    toc();
    if (r) return r;
    Attn128Args pa;
    pa.Q = qs_; pa.K = ks_; pa.O = Om_; pa.B = B; pa.L = P; pa.ldv = ldvP_; pa.split = split; pa.vcols = 32; pa.ldq = 256;
    pa.Vt = Vtf_; pa.v_bstride = (int64_t)64 * ldvP_;
    pa.q_bstride = pa.k_bstride = dirs == 2 ? img : 2 * img;
    if ((r = attention(pa, P))) return r;

    // ---- convex upsampling (gmflow.py:74-92), unpad, encode ----
    tic(F_ELT, 0, 0);
    r = launch_gm_upsampler_in(stream, Om_, X_, flowp_, umap_, B, P, dirs == 2 ? 1 : 2);
    toc();
    if (r) return r;
    fstages_["flow_prop"] = FStage{flowp_, B, P, 2};
    if ((r = conv(umap_, 192, 384, B, h8_, w8_, 3, 3, 1, up0_, u1_, 512, ACT_RELU, 0, nullptr, nullptr, split_w_ ? 256 : 0))) return r;
    if ((r = gemm32(u1_, 512, (int64_t)B * P, up2_, gmask_, 576))) return r;
    float *up = flow_out ? flow_out : gup_;
    tic(F_PP, 0, (double)B * sh_ * sw_ * 8);
    if (isz_h_ > 0) {                 // convex upsampling at the inference size, then the bilinear resize back to the scaled frame
        r = launch_upsample(stream, flowp_, gmask_, B, h8_, w8_, 0, 0, Hp_, Wp_, gupi_, gmaxd_);
        if (!r) r = launch_flow_resize_back(stream, gupi_, B, Hp_, Wp_, sh_, sw_, up, gmaxd_);
    } else {
        r = launch_upsample(stream, flowp_, gmask_, B, h8_, w8_, padl_, padt_, sh_, sw_, up, gmaxd_);
    }
    toc();
    if (r) return r;
    tic(F_PP, 0, (double)B * sh_ * sw_ * 11);
    r = launch_flow_encode(stream, up, B, sh_, sw_, gmaxd_, rgb_out, maxdisp);
    toc();
    if (r || !mask_out) return r;
    tic(F_PP, 0, (double)B * sh_ * sw_ * 17);
    r = launch_fwdbwd_mask(stream, up, NP, sh_, sw_, alpha1, alpha2, mask_out);
    toc();
    return r;
}

int64_t GmflowEngine::get_stage(const char *name, float *out, int64_t cap, int64_t shape[4]) {
    auto it = fstages_.find(name);
    PB_CHECK(it != fstages_.end(), PB_ERR_ARG, "unknown stage '%s'", name);
    const FStage &s = it->second;
    const int64_t total = s.n * s.rows * s.cols;
    PB_CHECK(total <= cap, PB_ERR_ARG, "stage buffer too small");
    PB_HIP(hipStreamSynchronize(stream));
    PB_HIP(hipMemcpy(out, s.ptr, (size_t)total * 4, hipMemcpyDeviceToHost));
    shape[0] = s.n; shape[1] = s.rows; shape[2] = s.cols; shape[3] = 1;
    return total;
}
