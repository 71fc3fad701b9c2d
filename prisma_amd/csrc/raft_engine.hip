// RaftEngine: the flow_raft band on one MI355X.
// Reference call stack being replaced: bands/flow_raft.py:98-113 (per-frame loop: resize, [prev,curr] /
// [curr,prev] batch, infer) -> :51-62 (pad, RAFT(test_mode), unpad) -> bands/raft/raft.py:87-146 ->
// extractor.py / corr.py / update.py -> bands/common/flow.py:64-88 + encode.py:98-126 (process_flow).
//
// Differences in schedule, not in arithmetic:
//   * every frame goes through fnet and cnet ONCE per sequence (the reference re-encodes both frames of
//     every pair, twice when it also computes the backward flow);
//   * the mask head (152.9 of 1559.6 GFLOP per pair at 720p) runs only after the last iteration - in
//     test_mode the reference computes it every iteration and discards all but the last (raft.py:136-144);
//   * cnet's eval-mode BatchNorm is folded into its convolutions at pack time; fnet's InstanceNorm uses
//     run-time statistics and stays a separate pass.
#include "raft_engine.h"

#include <math.h>
#include <string.h>

#include <algorithm>

namespace {
inline int cp64(int c) { return (int)round_up(c, 64); }

void cubic_taps_u8(int src, int dst, double scale, std::vector<int> &idx, std::vector<int> &co) {
    // OpenCV 8-bit INTER_CUBIC: float32 coefficients (A = -0.75) -> saturate_cast<short>(c * 2048)
    idx.resize((size_t)dst * 4);
    co.resize((size_t)dst * 4);
    const double inv = 1.0 / scale;
    const float A = -0.75f;
    for (int d = 0; d < dst; ++d) {
        float fx = (float)((d + 0.5) * inv - 0.5);
        const int sx = (int)floorf(fx);
        fx -= (float)sx;
        float c[4];
        c[0] = ((A * (fx + 1.f) - 5.f * A) * (fx + 1.f) + 8.f * A) * (fx + 1.f) - 4.f * A;
        c[1] = ((A + 2.f) * fx - (A + 3.f)) * fx * fx + 1.f;
        c[2] = ((A + 2.f) * (1.f - fx) - (A + 3.f)) * (1.f - fx) * (1.f - fx) + 1.f;
        c[3] = 1.f - c[0] - c[1] - c[2];
        for (int t = 0; t < 4; ++t) {
            idx[(size_t)d * 4 + t] = std::min(std::max(sx - 1 + t, 0), src - 1);
            co[(size_t)d * 4 + t] = (int)nearbyint((double)c[t] * 2048.0);
        }
    }
}
}  // namespace

void RaftEngine::out_size(int H, int W, float scale, int *sh, int *sw) {
    *sh = (int)nearbyint((double)H * scale);
    *sw = (int)nearbyint((double)W * scale);
}

// BasicEncoder weights (extractor.py:118-192) up to the last residual block: stem, layer1-3, downsample paths.  bnf: eval BatchNorm folded
// into the convolutions (cnet); conv_bias: the 3x3 / 7x7 convolutions carry biases (RAFT yes, GMFlow's CNNEncoder no - its 1x1 downsample
// convolutions do, backbone.py:22-25)
int RaftEngine::pack_encoder(const std::string &en, bool bnf, bool conv_bias, Enc &E) {
    int r;
    const int dims[3] = {64, 96, 128};
    std::vector<float> sc, sf;
    {   // stem 7x7 / stride 2 (extractor.py:124,171): on the 4 x 4 space-to-depth image (64 channels = (dy, dx, c), raft_prep) it is a
        // 3x3 / stride 1 convolution whose 4 x 64 output channels are the 2 x 2 output pixels of a block - an implicit GEMM on the
        // ping-pong kernel with the pixel-shuffle epilogue instead of an im2col round trip (2.8 GB per 32 frames at 1080p x 0.75).
        // W2[(sy, sx, o)][(ty, tx)][(dy, dx, c)] = w[o][c][ky][kx],  ky = 4 (ty - 1) + dy - 2 sy + 3 (same in x), zero outside 0..6
        auto iw = tmap_.find(en + ".conv1.weight"), ib = tmap_.find(en + ".conv1.bias");
        PB_CHECK(iw != tmap_.end() && (ib != tmap_.end() || !conv_bias), PB_ERR_ARG, "missing %s.conv1", en.c_str());
        if (bnf && (r = fold_bn(en + ".norm1", 64, sc, sf))) return r;
        const float *wt = (const float *)iw->second->data, *bs = conv_bias ? (const float *)ib->second->data : nullptr;
        std::vector<float> g((size_t)256 * 576, 0.f), bb(256);
        for (int sy = 0; sy < 2; ++sy)
            for (int sx = 0; sx < 2; ++sx)
                for (int o = 0; o < 64; ++o) {
                    const float s = bnf ? sc[o] : 1.f;
                    const int n = (sy * 2 + sx) * 64 + o;
                    bb[n] = (bs ? bs[o] : 0.f) * s + (bnf ? sf[o] : 0.f);
                    for (int ty = 0; ty < 3; ++ty)
                        for (int tx = 0; tx < 3; ++tx)
                            for (int dy = 0; dy < 4; ++dy)
                                for (int dx = 0; dx < 4; ++dx) {
                                    const int ky = 4 * (ty - 1) + dy - 2 * sy + 3, kx = 4 * (tx - 1) + dx - 2 * sx + 3;
                                    if (ky < 0 || ky > 6 || kx < 0 || kx > 6) continue;
                                    for (int c = 0; c < 3; ++c)
                                        g[(size_t)n * 576 + (ty * 3 + tx) * 64 + (dy * 4 + dx) * 4 + c] = wt[((size_t)o * 3 + c) * 49 + ky * 7 + kx] * s;
                                }
                }
        if ((r = pack(g.data(), 256, 576, 576, E.stem, bb.data(), 9, 1))) return r;
        E.stem.Kreal = 147;
    }
    for (int li = 0; li < 3; ++li)
        for (int bi = 0; bi < 2; ++bi) {
            const std::string p = en + ".layer" + std::to_string(li + 1) + "." + std::to_string(bi);
            for (int c = 0; c < 2; ++c) {
                if (bnf && (r = fold_bn(p + ".norm" + std::to_string(c + 1), dims[li], sc, sf))) return r;
                if ((r = pack_conv(p + ".conv" + std::to_string(c + 1), conv_bias, bnf ? sc.data() : nullptr, bnf ? sf.data() : nullptr,
                                   E.l[li][bi][c], 1)))
                    return r;
            }
            if (bi == 0 && li > 0) {
                if (bnf && (r = fold_bn(p + ".norm3", dims[li], sc, sf))) return r;
                if ((r = pack_conv(p + ".downsample.0", true, bnf ? sc.data() : nullptr, bnf ? sf.data() : nullptr, E.ds[li], 1))) return r;
            }
        }
    return 0;
}

int RaftEngine::load(const pb_tensor *w, int n) {
    int r0 = begin_load(w, n);
    if (r0) return r0;
    int r;
    pack_tapin_ = getenv("PB_TAPIN") && getenv("PB_TAPIN")[0] == '2';
    for (int e = 0; e < 2; ++e) {
        const std::string en = e == 0 ? "fnet" : "cnet";
        Enc &E = e == 0 ? fnet_ : cnet_;
        if ((r = pack_encoder(en, e == 1, true, E))) return r;
        if ((r = pack_conv(en + ".conv2", true, nullptr, nullptr, E.out, 1))) return r;
    }
    const std::string u = "update_block.";
    // the update block's activations carry an fp8 copy ([a16 | a8] per pixel) and its weight residuals are e4m3 (PackedW::mx2): the residual
    // pass a8 x w_lo8 runs on the MX-scaled MFMA at half the matrix-pipe time of the second fp16 pass it replaces.  History: rounds 2-3 kept
    // this off (it bought 1 % of the band: the MX build of the 128 x 128 tile copied its accumulators through VGPRs every K tile, and the
    // variant excluded the context hoist below); round 4 fixed both - 31 pairs 1080p x 0.75: flow band 150.4 -> 142.2 ms on one box - and the
    // parity figures do not move (720p x 8 pairs 4.1e-4 ... 5.4e-4 max, 3.4e-4 ... 3.5e-4 L2 against 4.1e-4 ... 6.6e-4 / 3.4e-4 with two fp16
    // passes; heavy-tailed weights 5.8e-4 against 6.3e-4): the e4m3 rounding only ever enters a residual term.  PB_MX_UPD=0: two fp16 passes.
    pack_mx2_ = mx_ && !(getenv("PB_MX_UPD") && getenv("PB_MX_UPD")[0] == '0');
    upd8_ = pack_mx2_;
    // slice-major K order (gemm.h cTapInner) for the 3x3 / 1x5 / 5x1 convolutions over the 128 ... 384-channel maps of the update block:
    // tap-major order overflows the XCD L2 there.  PB_TAPIN=0 turns it off, 2 also applies it to the encoders (A/B runs).
    const char *tin = getenv("PB_TAPIN");
    const int tapin_all = pack_tapin_;
    pack_tapin_ = !(tin && tin[0] == '0');
    if ((r = pack_conv(u + "encoder.convc1", true, nullptr, nullptr, convc1_))) return r;
    if ((r = pack_conv(u + "encoder.convc2", true, nullptr, nullptr, convc2_))) return r;
    if ((r = pack_conv(u + "encoder.convf2", true, nullptr, nullptr, convf2_))) return r;
    if ((r = pack_conv(u + "encoder.conv", true, nullptr, nullptr, convm_))) return r;
    convm_.N = 128; convm_.Nreal = 126;                     // 126 real outputs + 2 zero rows (N must be a multiple of 8)
    {   // convf1 7x7 on the 2-channel flow: im2col order k = tap*2 + c, K 98 -> 128
        auto iw = tmap_.find(u + "encoder.convf1.weight"), ib = tmap_.find(u + "encoder.convf1.bias");
        PB_CHECK(iw != tmap_.end() && ib != tmap_.end(), PB_ERR_ARG, "missing convf1");
        const float *wt = (const float *)iw->second->data;
        std::vector<float> g((size_t)128 * 98);
        for (int o = 0; o < 128; ++o)
            for (int c = 0; c < 2; ++c)
                for (int tp = 0; tp < 49; ++tp) g[(size_t)o * 98 + tp * 2 + c] = wt[((size_t)o * 2 + c) * 49 + tp];
        if ((r = pack(g.data(), 128, 98, 128, convf1_, (const float *)ib->second->data))) return r;
        {
            const char *e = getenv("PB_CONVF1_DIRECT");
            f1_direct_ = !(e && e[0] == '0');
            f1_passes_ = split_w_ ? 2 : 1;
            std::vector<f16> hw((size_t)convf1_packed_halfs(f1_passes_));
            convf1_pack(wt, f1_passes_, hw.data());
            void *pw = nullptr;
            PB_HIP(hipMalloc(&pw, hw.size() * 2));
            owned_.push_back(pw);
            PB_HIP(hipMemcpy(pw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
            f1w_ = (f16 *)pw;
        }
    }
    // Context hoist (default; PB_GRU_HOIST=0 turns it off): the GRU's input is cat(h, inp, motion) and `inp` - the context
    // features - does not change over the iterations (raft.py:112-115, update.py:131).  A convolution is linear in its input channels, so
    // the inp share of every gate's pre-activation is computed ONCE per call (bias-free, stored as an fp16 hi plane + an fp16 lo plane: ~22
    // bits) and the per-iteration convolutions run on [h | motion] alone (K = 5 x 256 instead of 5 x 384), adding the two planes as the
    // epilogue's skip tensors (add1, add2) before the gate's activation.  The buffers hold [h | motion | inp] so that the two varying
    // parts are adjacent.  (First version: fp32 share as the accumulators' starting value - its 256 KB per tile were a prologue burst that
    // ate half of what the smaller K saved; as an fp32 read in the GRU epilogues it cost every EPI_STD kernel its register allocation.)
    {
        const char *e = getenv("PB_GRU_HOIST");
        hoist_ = !(e && e[0] == '0');
    }
    for (int half = 0; half < 2; ++half) {
        // z and r gates share their input: one GEMM with N = 256 ([z | r]); q separately
        const std::string sfx = std::to_string(half + 1);
        auto get = [&](const std::string &nm) -> const pb_tensor * {
            auto it = tmap_.find(nm);
            return it == tmap_.end() ? nullptr : it->second;
        };
        const pb_tensor *wz = get(u + "gru.convz" + sfx + ".weight"), *wr = get(u + "gru.convr" + sfx + ".weight");
        const pb_tensor *bz = get(u + "gru.convz" + sfx + ".bias"), *br = get(u + "gru.convr" + sfx + ".bias");
        const pb_tensor *wq = get(u + "gru.convq" + sfx + ".weight"), *bq = get(u + "gru.convq" + sfx + ".bias");
        PB_CHECK(wz && wr && bz && br && wq && bq, PB_ERR_ARG, "missing gru weights");
        // rows [o][tap][c] over the channel range [c0, c0 + nc) of the reference's 384 input channels (h 0..127, inp 128..255, motion 256..383),
        // written at column offset `dst` of a [taps x width] row
        auto fill = [&](std::vector<float> &g, int row0, int width, int dst, const pb_tensor *w, int c0, int nc) {
            const float *wt = (const float *)w->data;
            for (int o = 0; o < 128; ++o)
                for (int c = 0; c < nc; ++c)
                    for (int tp = 0; tp < 5; ++tp) g[(size_t)(row0 + o) * 5 * width + tp * width + dst + c] = wt[((size_t)o * 384 + c0 + c) * 5 + tp];
        };
        std::vector<float> bb(256);
        for (int o = 0; o < 128; ++o) { bb[o] = ((const float *)bz->data)[o]; bb[128 + o] = ((const float *)br->data)[o]; }
        if (!hoist_) {
            const int K = 5 * 384;
            std::vector<float> g((size_t)256 * K, 0.f);
            fill(g, 0, 384, 0, wz, 0, 384);
            fill(g, 128, 384, 0, wr, 0, 384);
            if ((r = pack(g.data(), 256, K, K, zr_[half], bb.data(), 5))) return r;
            if ((r = pack_conv(u + "gru.convq" + sfx, true, nullptr, nullptr, q_[half]))) return r;
            continue;
        }
        {   // [h | motion] parts (with the biases) and the inp parts (bias-free: their result is the accumulators' starting value)
            std::vector<float> g((size_t)256 * 5 * 256, 0.f), gi((size_t)256 * 5 * 128, 0.f);
            fill(g, 0, 256, 0, wz, 0, 128);   fill(g, 0, 256, 128, wz, 256, 128);
            fill(g, 128, 256, 0, wr, 0, 128); fill(g, 128, 256, 128, wr, 256, 128);
            fill(gi, 0, 128, 0, wz, 128, 128);
            fill(gi, 128, 128, 0, wr, 128, 128);
            if ((r = pack(g.data(), 256, 5 * 256, 5 * 256, zr_[half], bb.data(), 5))) return r;
            pack_mx2_ = 0;                                  // the once-per-call share runs two fp16 passes into an fp16 hi + lo plane
            r = pack(gi.data(), 256, 5 * 128, 5 * 128, zr_in_[half], nullptr, 5);
            pack_mx2_ = upd8_;
            if (r) return r;
        }
        {
            std::vector<float> g((size_t)128 * 5 * 256, 0.f), gi((size_t)128 * 5 * 128, 0.f);
            fill(g, 0, 256, 0, wq, 0, 128); fill(g, 0, 256, 128, wq, 256, 128);
            fill(gi, 0, 128, 0, wq, 128, 128);
            if ((r = pack(g.data(), 128, 5 * 256, 5 * 256, q_[half], (const float *)bq->data, 5))) return r;
            pack_mx2_ = 0;
            r = pack(gi.data(), 128, 5 * 128, 5 * 128, q_in_[half], nullptr, 5);
            pack_mx2_ = upd8_;
            if (r) return r;
        }
    }
    if ((r = pack_conv(u + "flow_head.conv1", true, nullptr, nullptr, fh1_))) return r;
    pack_mx2_ = 0;                                          // flow_head2 is a direct kernel, mask.2 an fp32-output GEMM: fp16 residuals
    const int tapin_upd = pack_tapin_;
    pack_tapin_ = 0;                                        // ... that walks its weights tap-major
    if ((r = pack_conv(u + "flow_head.conv2", true, nullptr, nullptr, fh2_))) return r;
    fh2_.N = 8;                                             // 2 real outputs, rows 2..7 are zero
    pack_mx2_ = upd8_;
    pack_tapin_ = tapin_upd;
    if ((r = pack_conv(u + "mask.0", true, nullptr, nullptr, mk0_))) return r;
    pack_mx2_ = 0;
    pack_tapin_ = tapin_all;
    if ((r = pack_conv(u + "mask.2", true, nullptr, nullptr, mk2_))) return r;
    tmap_.clear();
    PB_HIP(hipDeviceSynchronize());
    return 0;
}

int RaftEngine::prepare(int F, int H, int W, float scale, int dirs) {
    if (F <= pF_ && H == pH_ && W == pW_ && scale == pS_ && dirs <= pD_) return 0;
    PB_HIP(hipStreamSynchronize(stream));
    // from here on the plan's members are being rewritten: a failure below (a frame too small, an allocation) must not leave the OLD plan's key
    // behind, or the next call with the old size would skip this function and run on the half-written geometry (round 6: found by
    // tests/test_gpu_raft.py::test_pipeline_error_exit_leaves_no_copy_in_flight - wrong flows after a refused 40 x 40 call)
    pF_ = 0; pH_ = 0; pW_ = 0; pD_ = 0;
    geometry(H, W, scale, 8);
    PB_CHECK(h8_ >= 16 && w8_ >= 16, PB_ERR_ARG, "flow_raft: %dx%d is too small (the 4-level pyramid needs >= 128 px)", sh_, sw_);
    P8_ = (P_ + 7) / 8 * 8;       // row stride of the level-0 volume (the GEMM epilogue writes 8-column groups)
    lh_[0] = h8_; lw_[0] = w8_;
    for (int l = 1; l < 4; ++l) { lh_[l] = lh_[l - 1] / 2; lw_[l] = lw_[l - 1] / 2; }
    const int64_t ND = (int64_t)(F - 1) * dirs;
    const size_t slack = 1 << 20;
    for (int pass = 0; pass < 2; ++pass) {
        planning_ = pass == 0;
        arena_off_ = 0;
        carve_encoder(F);
        fmap_ = (f16 *)carve((size_t)round_up((int64_t)F * P_, 256) * 256 * 2 + slack);
        ctx_ = (f16 *)carve((size_t)round_up((int64_t)F * P_, 256) * 256 * 2);
        for (int l = 0; l < 4; ++l) {
            // level l of the volume: fp16, one row per source pixel, targets in 8 x 8 tiles (raft_kernels.hip corr_tile_kernel)
            lwp_[l] = (int)round_up(lw_[l], 8);
            pld_[l] = (int)round_up(lh_[l], 8) * lwp_[l];
            // a row stride that is a multiple of 256 lets launch_gemm pick the 256 x 256 ping-pong kernel for the level (measured at
            // 18360 x 19136 x 256: 0.410 -> 0.339 ms, tools/volume_gemm_probe.py); taken when it costs under 2 % of the level's bytes
            if (round_up(pld_[l], 256) * 50 <= (int64_t)pld_[l] * 51) pld_[l] = (int)round_up(pld_[l], 256);
            pyr_[l] = (f16 *)carve((size_t)ND * P_ * pld_[l] * 2 + slack);
            fpool_[l] = l == 0 ? nullptr : (f16 *)carve((size_t)round_up((int64_t)F * lh_[l] * lw_[l], 256) * 256 * 2 + slack);
            ftile_[l] = (f16 *)carve((size_t)(F * (int64_t)pld_[l] + 256) * 256 * 2 + slack);
        }
        const int64_t rows = round_up(ND * P_, 256);
        h32_ = (float *)carve((size_t)rows * 128 * 4); flow_ = (float *)carve((size_t)rows * 2 * 4);
        mask_ = (float *)carve((size_t)rows * 576 * 4);
        const int Lq = upd8_ ? 576 : 384;
        const size_t u8 = upd8_ ? 3 : 2;                   // bytes per channel of an update-block map: fp16 (+ its fp8 copy after the pixel's fp16 part)
        hx_ = (f16 *)carve((size_t)rows * 384 * u8); hx2_ = (f16 *)carve((size_t)rows * 384 * u8);
        corr_ = (f16 *)carve((size_t)rows * 384 * u8); c1_ = (f16 *)carve((size_t)rows * 256 * u8);
        corflo_ = (f16 *)carve((size_t)rows * 256 * u8); fa_ = (f16 *)carve((size_t)rows * 128 * u8);
        f1_ = (f16 *)carve((size_t)rows * 128 * u8); zrb_ = (f16 *)carve((size_t)rows * 256 * 2);
        fh_ = (f16 *)carve((size_t)rows * 256 * 2);
        for (int half = 0; half < 2; ++half) {            // the context features' share of the GRU pre-activations (hoist_): hi plane, lo plane
            gz_[half] = hoist_ ? (f16 *)carve((size_t)rows * 256 * 2 * 2) : nullptr;           // row stride 256 = zrb_'s
            gq_[half] = hoist_ ? (f16 *)carve((size_t)rows * Lq * 2 * 2) : nullptr;            // row stride = hx_'s (the q convolution writes h there)
        }
        m0_ = (f16 *)carve((size_t)rows * 256 * 2);
        up_ = (float *)carve((size_t)ND * sh_ * sw_ * 2 * 4);
        maxd_ = (unsigned *)carve((size_t)ND * 4);
        if (pass == 0) {
            const int rc = commit_arena("flow");
            if (rc) return rc;
        }
    }
    int rt = upload_resize_tables(H, W, scale);
    if (rt) return rt;
    pF_ = F; pH_ = H; pW_ = W; pS_ = scale; pD_ = dirs;
    return 0;
}

// scaled size, InputPadder('sintel', padding_factor = factor) amounts (bands/common/flow.py:43-55) and the 1/8 grid
void RaftEngine::geometry(int H, int W, float scale, int factor) {
    out_size(H, W, scale, &sh_, &sw_);
    const int ph = (((sh_ / factor) + 1) * factor - sh_) % factor, pw = (((sw_ / factor) + 1) * factor - sw_) % factor;
    padl_ = pw / 2; padt_ = ph / 2;
    Hp_ = sh_ + ph; Wp_ = sw_ + pw;
    h8_ = Hp_ / 8; w8_ = Wp_ / 8; P_ = h8_ * w8_;
}

// arena buffers of frame prep and run_encoder (call between the planning / carving passes of a prepare)
void RaftEngine::carve_encoder(int F) {
    const int h2 = Hp_ / 2, w2 = Wp_ / 2, h4 = Hp_ / 4, w4 = Wp_ / 4;
    xi_ = (int *)carve((size_t)sw_ * 16); xc_ = (int *)carve((size_t)sw_ * 16);
    yi_ = (int *)carve((size_t)sh_ * 16); yc_ = (int *)carve((size_t)sh_ * 16);
    // split-fp16 mode: the encoders' maps (image included) are [hi | lo] per pixel (es = 2)
    const size_t es = split_w_ ? 2 : 1;
    img_ = (f16 *)carve((size_t)F * Hp_ * Wp_ * 8 * es);
    for (auto &b : r1_) b = (f16 *)carve((size_t)round_up((int64_t)F * h2 * w2, 256) * 64 * 2 * es);
    for (auto &b : r2_) b = (f16 *)carve((size_t)round_up((int64_t)F * h4 * w4, 256) * 128 * 2 * es);
    for (auto &b : r3_) b = (f16 *)carve((size_t)round_up((int64_t)F * P_, 256) * 128 * 2 * es);
    for (auto &b : st_) b = (float *)carve((size_t)F * 256 * 2 * 4);
    stp_ = (float *)carve((size_t)in_stats_chunks(h2 * w2) * F * 256 * 2 * 4);     // per-chunk partial sums of the largest map
}

int RaftEngine::upload_resize_tables(int H, int W, float scale) {
    if (scale != 1.f) {
        std::vector<int> xi, xc, yi, yc;
        cubic_taps_u8(W, sw_, scale, xi, xc);
        cubic_taps_u8(H, sh_, scale, yi, yc);
        PB_HIP(hipMemcpyAsync(xi_, xi.data(), xi.size() * 4, hipMemcpyHostToDevice, stream));
        PB_HIP(hipMemcpyAsync(xc_, xc.data(), xc.size() * 4, hipMemcpyHostToDevice, stream));
        PB_HIP(hipMemcpyAsync(yi_, yi.data(), yi.size() * 4, hipMemcpyHostToDevice, stream));
        PB_HIP(hipMemcpyAsync(yc_, yc.data(), yc.size() * 4, hipMemcpyHostToDevice, stream));
    }
    PB_HIP(hipStreamSynchronize(stream));
    return 0;
}

// BasicEncoder.forward (extractor.py:171-192) without its last 1x1 convolution, on the F prepared frames in img_: stem, three stages of two
// residual blocks.  inorm: InstanceNorm with run-time statistics (fnet; GMFlow's backbone) - otherwise the folded BatchNorm path (cnet).
// *x_out = the last block's output map [F, h8, w8, 128] (split-fp16 layout in PB_PREC_SPLIT).
int RaftEngine::run_encoder(const Enc &E, bool inorm, int F, const f16 **x_out) {
    int r = 0;
    const int h2 = Hp_ / 2, w2 = Wp_ / 2;
    const int es = split_w_ ? 2 : 1;                         // encoder maps are [hi | lo] in split-fp16 mode
    auto lo = [&](int c) { return split_w_ ? c : 0; };
    const int l8 = split_w_ && mx_ ? kLo8Pa : -1;            // the residual parts of the encoder maps are e4m3 ([hi | hi8 | lo8])
    auto norm_relu = [&](const f16 *t, float *st, f16 *y, int HW, int C, const f16 *b, const float *sb) -> int {
        tic(F_ELT, 0, 0);
        int rr = launch_in_apply(stream, t, st, b, sb, y, F, HW, C, es * C, lo(C), l8);
        toc();
        return rr;
    };
    // On the e4m3-residual maps (flow_raft) the instance-norm statistics are taken from the hi parts alone (PB_IN_STATS_LO=1: from hi + lo8
    // as before): the fp16 rounding residuals are symmetric around zero, so over a channel's pixels they move the mean by ~2^-12 / sqrt(n) of a
    // value, and a 64-channel pixel's hi part is one of its two 128-byte lines (the lo8 bytes sit in the other one): half the pass's traffic,
    // -1.2 ms per step, RAFT parity unchanged (worst 720p pair 6.0e-4 either way).  flow_gmflow (fp16 residual planes) keeps hi + lo: its
    // 216x300 vector moves from 2.2e-4 to 3.8e-4 without them.
    static const bool stats_lo_env = getenv("PB_IN_STATS_LO") && getenv("PB_IN_STATS_LO")[0] == '1';
    const bool stats_lo = stats_lo_env || !mx_;
    auto stats = [&](const f16 *t, float *st, int HW, int C) -> int {
        tic(F_ELT, 0, 0);
        int rr = launch_in_stats(stream, t, F, HW, C, es * C, stp_, st, stats_lo ? lo(C) : 0, l8);
        toc();
        return rr;
    };
    // stem (BasicEncoder.forward, extractor.py:171-192)
    {
        GemmArgs a;
        a.A = img_; a.N = 256;
        a.cH = Hp_ / 4; a.cW = Wp_ / 4; a.cC = 64; a.cLd = es * 64; a.cKW = 3; a.cStride = 1; a.cPad = 1; a.cPadX = 1;
        set_weights(a, E.stem, true);
        a.cOH = a.cH; a.cOW = a.cW; a.M = F * a.cH * a.cW;
        a.out = r1_[5]; a.ldo = es * 64; a.lo_off = lo(64); a.act = inorm ? ACT_NONE : ACT_RELU;
        if (l8 >= 0) { a.lo8 = 1; a.lo8_pa = l8; }
        a.ps_h = a.cH; a.ps_w = a.cW; a.ps_s = 2; a.ps_co = 64;
        tic(F_CONV, 2.0 * F * h2 * w2 * 64.0 * 147, 0, E.stem.mx3 ? 2.0 : 1.0 + E.stem.sa + E.stem.sw);
        r = launch_gemm(stream, A_CONV, EPI_PIXSHUF, TILE_AUTO, a);
        if (timer.enabled && !r && !open_.empty()) timer.recs[open_.back()].name = pb_gemm_last_kernel();
        toc();
        if (r) return r;
    }
    const f16 *x = r1_[5];
    if (inorm) {
        if ((r = stats(r1_[5], st_[0], h2 * w2, 64))) return r;
        if ((r = norm_relu(r1_[5], st_[0], r1_[6], h2 * w2, 64, nullptr, nullptr))) return r;
        x = r1_[6];
    }
    int H_ = h2, W_ = w2, C_ = 64;
    for (int li = 0; li < 3; ++li) {
        const int stride = li == 0 ? 1 : 2;
        const int Cn = li == 0 ? 64 : 128;               // 96 is carried as 128 (zero padded channels)
        f16 **R = li == 0 ? r1_ : (li == 1 ? r2_ : r3_);
        for (int bi = 0; bi < 2; ++bi) {                  // ResidualBlock.forward (extractor.py:46-56)
            const int s = bi == 0 ? stride : 1;
            const int OH = (H_ - 1) / s + 1, OW = (W_ - 1) / s + 1;
            const int Cin = bi == 0 ? C_ : Cn;
            f16 *t1 = R[0], *t2 = R[1], *t3 = R[2], *outb = R[3 + bi];
            if (inorm) {
                if ((r = conv(x, Cin, es * Cin, F, H_, W_, 3, 3, s, E.l[li][bi][0], t1, es * Cn, ACT_NONE, 0, nullptr, nullptr, lo(Cn)))) return r;
                if ((r = stats(t1, st_[0], OH * OW, Cn))) return r;
                if ((r = norm_relu(t1, st_[0], t1, OH * OW, Cn, nullptr, nullptr))) return r;
                if ((r = conv(t1, Cn, es * Cn, F, OH, OW, 3, 3, 1, E.l[li][bi][1], t2, es * Cn, ACT_NONE, 0, nullptr, nullptr, lo(Cn)))) return r;
                if ((r = stats(t2, st_[1], OH * OW, Cn))) return r;
                if (s != 1) {
                    if ((r = conv(x, Cin, es * Cin, F, H_, W_, 1, 1, s, E.ds[li], t3, es * Cn, ACT_NONE, 0, nullptr, nullptr, lo(Cn)))) return r;
                    if ((r = stats(t3, st_[2], OH * OW, Cn))) return r;
                    if ((r = norm_relu(t2, st_[1], outb, OH * OW, Cn, t3, st_[2]))) return r;
                } else {
                    if ((r = norm_relu(t2, st_[1], outb, OH * OW, Cn, x, nullptr))) return r;
                }
            } else {
                if ((r = conv(x, Cin, es * Cin, F, H_, W_, 3, 3, s, E.l[li][bi][0], t1, es * Cn, ACT_RELU, 0, nullptr, nullptr, lo(Cn)))) return r;
                const f16 *xs = x;
                if (s != 1) {
                    if ((r = conv(x, Cin, es * Cin, F, H_, W_, 1, 1, s, E.ds[li], t3, es * Cn, ACT_NONE, 0, nullptr, nullptr, lo(Cn)))) return r;
                    xs = t3;
                }
                if ((r = conv(t1, Cn, es * Cn, F, OH, OW, 3, 3, 1, E.l[li][bi][1], outb, es * Cn, ACT_RELU, 1, xs, nullptr, lo(Cn)))) return r;
            }
            x = outb; H_ = OH; W_ = OW; C_ = Cn;
        }
    }
    *x_out = x;
    return 0;
}

int RaftEngine::infer(const uint8_t *frames, int F, int H, int W, float scale, int iters, int backward, float *flow_out,
                      uint8_t *rgb_out, float *maxdisp, uint8_t *mask_out, float alpha1, float alpha2) {
    PB_CHECK(frames && F >= 2 && H > 0 && W > 0 && iters >= 1 && scale > 0.f, PB_ERR_ARG, "flow infer: bad arguments");
    PB_CHECK(!mask_out || backward, PB_ERR_ARG, "consistency masks need both directions (backward = 1)");
    PB_HIP(hipSetDevice(device));
    const int dirs = backward ? 2 : 1;
    int r = prepare(F, H, W, scale, dirs);
    if (r) return r;
    timer.reset();
    stages_.clear();
    const int ND = (F - 1) * dirs;
    last_nd_ = ND;
    const int64_t rows = (int64_t)ND * P_;

    // ---- frame prep + stem im2col (shared by fnet and cnet) ----
    tic(F_PP, 0, (double)F * H * W * 3);
    const int Lhx = upd8_ ? 576 : 384, L256 = upd8_ ? 384 : 256, L128 = upd8_ ? 192 : 128;    // pixel strides of the update block's maps
    const float s8 = (float)(1 << kMx2Pa);
    const int es = split_w_ ? 2 : 1;                         // encoder maps are [hi | lo] in split-fp16 mode
    auto lo = [&](int c) { return split_w_ ? c : 0; };
    const int l8 = split_w_ && mx_ ? kLo8Pa : -1;            // the residual parts of the encoder maps are e4m3 ([hi | hi8 | lo8])
    r = launch_raft_prep(stream, frames, F, H, W, sh_, sw_, Hp_, Wp_, padl_, padt_, scale != 1.f, xi_, xc_, yi_, yc_, img_, nullptr, 1, lo(64), l8);
    toc();
    if (r) return r;

    // ---- the two encoders ----
    for (int e = 0; e < 2; ++e) {
        const Enc &E = e == 0 ? fnet_ : cnet_;
        const f16 *x = nullptr;
        // the context network runs on the SOURCE frame of every pair (raft.py:110-115): frames 0 .. F - 2 forward, 1 .. F - 1 backward - without the
        // backward direction the clip's last frame needs no context features (1 / F of cnet's work; per-frame results do not depend on the batch)
        const int Fe = e == 1 && dirs == 1 ? F - 1 : F;
        if ((r = run_encoder(E, e == 0, Fe, &x))) return r;
        if ((r = dense(x, es * 128, (int64_t)Fe * P_, E.out, e == 0 ? fmap_ : ctx_, 256, ACT_NONE))) return r;
    }
    stages_["fmap"] = Stage{fmap_, 1, 0, 256, h8_, w8_, 256, 0};

    // ---- all-pairs correlation pyramid, recurrent state ----
    // corr.py:22-27 pools the volume over the target dims; pooling is linear, so level l is the correlation of fmap1 with the
    // l-times avg-pooled target features: three small pooling passes over [F, P, 256] instead of three over the 4 P^2-byte volume
    for (int l = 0; l < 4; ++l) {
        tic(F_ELT, 0, 0);
        if (l > 0) r = launch_avgpool2_nhwc(stream, l == 1 ? fmap_ : fpool_[l - 1], fpool_[l], F, lh_[l - 1], lw_[l - 1], 256);
        if (!r) r = launch_corr_tile(stream, l == 0 ? fmap_ : fpool_[l], ftile_[l], F, lh_[l], lw_[l], lwp_[l], pld_[l]);
        toc();
        if (r) return r;
    }
    for (int i = 0; i < F - 1; ++i)
        for (int d = 0; d < dirs; ++d) {
            const int n = i * dirs + d;
            for (int l = 0; l < 4; ++l) {
                GemmArgs a;
                a.A = fmap_ + (int64_t)(i + d) * P_ * 256; a.lda = 256; a.M = P_;
                a.W = ftile_[l] + (int64_t)(i + 1 - d) * pld_[l] * 256;
                a.K = 256; a.N = pld_[l];
                a.out = pyr_[l] + (int64_t)n * P_ * pld_[l]; a.ldo = pld_[l]; a.zero = zero_;
                // K = 256 against P x pld fp16 outputs: this launch is bound by writing the volume, not by the matrix pipe (bytes: A + W + out)
                tic(F_GEMM, 2.0 * P_ * (double)lh_[l] * lw_[l] * 256, 2.0 * ((double)P_ * 256 + (double)pld_[l] * 256 + (double)P_ * pld_[l]));
                static const bool generic = getenv("PB_VOLUME") && getenv("PB_VOLUME")[0] == '0';     // A/B: the generic GEMM kernels
                if (generic) {
                    r = launch_gemm(stream, A_DENSE, EPI_STD, TILE_AUTO, a);
                    if (timer.enabled && !r && !open_.empty()) timer.recs[open_.back()].name = pb_gemm_last_kernel();
                } else {        // volume.hip: A-stationary, persistent along the targets; same accumulation order, same bits
                    r = launch_corr_volume(stream, a.A, a.M, a.W, a.N, a.N, a.out, a.ldo);
                    if (timer.enabled && !r && !open_.empty()) timer.recs[open_.back()].name = "corr_volume_kernel";
                }
                toc();
                if (r) return r;
            }
            tic(F_ELT, 0, 0);
            r = launch_init_state(stream, ctx_ + (int64_t)(i + d) * P_ * 256, h32_ + (int64_t)n * P_ * 128,
                                  hx_ + (int64_t)n * P_ * Lhx, hx2_ + (int64_t)n * P_ * Lhx, flow_ + (int64_t)n * P_ * 2, P_, Lhx, upd8_ ? 768 : 0, s8, hoist_ ? 256 : 128);
            toc();
            if (r) return r;
        }

    if (debug && (r = snapshot("net0", h32_, (size_t)ND * P_ * 128 * 4, Stage{nullptr, 0, 0, P_, 1, 128, 128, (int64_t)P_ * 128}))) return r;

    // ---- the context features' share of the SepConvGRU gates, once per call (hoist_, see load()) ----
    const int mot = hoist_ ? 128 : 256;                     // channel offset of the motion features inside hx_ / hx2_
    if (hoist_) {
        PB_CHECK(rows * Lhx < (1LL << 31), PB_ERR_ARG, "flow_raft: %lld update-block rows exceed the 32-bit plane offset of the hoisted GRU share", (long long)rows);
        const int mx_saved = mx_;
        mx_ = 0;                                             // fp16 residual plane (these weights are packed without MX tiles)
        for (int half = 0; half < 2; ++half) {
            const int kh = half == 0 ? 1 : 5, kw = half == 0 ? 5 : 1;
            // [hi plane | lo plane]: lo_off = one whole plane, so both planes have the row stride of the tensor they are added to
            if ((r = conv(hx_ + 256, 128, Lhx, ND, h8_, w8_, kh, kw, 1, zr_in_[half], gz_[half], 256, ACT_NONE, 0, nullptr, nullptr, (int)(rows * 256)))) break;
            if ((r = conv(hx_ + 256, 128, Lhx, ND, h8_, w8_, kh, kw, 1, q_in_[half], gq_[half], Lhx, ACT_NONE, 0, nullptr, nullptr, (int)(rows * Lhx)))) break;
        }
        mx_ = mx_saved;
        if (r) return r;
    }

    // ---- GRU iterations (raft.py:124-144, update.py:122-136) ----
    for (int it = 0; it < iters; ++it) {
        tic(F_ELT, 0, 0);
        r = launch_corr_lookup(stream, pyr_, lh_, lw_, lwp_, pld_, flow_, P_, w8_, corr_, rows, Lhx, upd8_ ? 768 : 0, s8);
        toc();
        if (r) return r;
        if (debug && it == 0 && (r = snapshot("corr0", corr_, (size_t)ND * P_ * Lhx * 2, Stage{nullptr, 1, ND, 324, h8_, w8_, Lhx, 0}))) return r;
        // BasicMotionEncoder.  With upd8_ every map is [a16 (C) | a8 (C bytes)] per pixel (pixel stride 1.5 C halfs): the conv epilogues
        // store the fp8 copy too (o8 = its byte offset from the output row: 2 Ctot - slice offset) and the MX segments read it
        if ((r = conv(corr_, 384, Lhx, ND, h8_, w8_, 1, 1, 1, convc1_, c1_, L256, ACT_RELU, 0, nullptr, nullptr, 0, 0, upd8_ ? 512 : 0))) return r;
        if ((r = conv(c1_, 256, L256, ND, h8_, w8_, 3, 3, 1, convc2_, corflo_, L256, ACT_RELU, 0, nullptr, nullptr, 0, 0, upd8_ ? 512 : 0))) return r;
        if (f1_direct_) {
            tic(F_CONV, 2.0 * rows * 128.0 * 98.0, 8.0 * rows + 256.0 * rows, (double)f1_passes_);
            r = launch_convf1(stream, flow_, f1w_, convf1_.bias, f1_, rows, P_, h8_, w8_, L128, upd8_ ? 256 : 0, (float)(1 << kMx2Pa), f1_passes_);
            if (timer.enabled && !r) timer.recs[open_.back()].name = "convf1_kernel";
            toc();
            if (r) return r;
        } else {
            tic(F_ELT, 0, 0);
            r = launch_im2col7_flow(stream, flow_, ND, h8_, w8_, fa_, 128, L128, upd8_);
            toc();
            if (r) return r;
            if ((r = dense(fa_, L128, rows, convf1_, f1_, L128, ACT_RELU, nullptr, upd8_ ? 256 : 0, 0))) return r;     // flow is in pixels: its fp8 copy is unscaled
        }
        if ((r = conv(f1_, 128, L128, ND, h8_, w8_, 3, 3, 1, convf2_, corflo_ + 192, L256, ACT_RELU, 0, nullptr, nullptr, 0, 0, upd8_ ? 512 - 192 : 0))) return r;
        // HX = [h | inp | motion] feeds the z / r convs, HX2 = [r * h | inp | motion] the q conv: the motion features are
        // written to both by the producing conv (its ReLU'd second output), r * h and the state update by the GRU epilogues
        ConvFuse dup; dup.out2 = hx2_ + mot;
        if ((r = conv(corflo_, 256, L256, ND, h8_, w8_, 3, 3, 1, convm_, hx_ + mot, Lhx, ACT_RELU, 0, nullptr, &dup, 0, 0, upd8_ ? 768 - mot : 0))) return r;
        tic(F_ELT, 0, 0);
        r = launch_put_flow(stream, flow_, hx_, hx2_, rows, Lhx, upd8_ ? 768 : 0, s8, mot + 126);
        toc();
        if (r) return r;
        // SepConvGRU: (1 x 5) then (5 x 1)
        for (int half = 0; half < 2; ++half) {
            const int kh = half == 0 ? 1 : 5, kw = half == 0 ? 5 : 1;
            const int gc = hoist_ ? 256 : 384;               // channels the per-iteration convolutions read: [h | motion] (+ inp without the hoist)
            ConvFuse fz; fz.gru_h = h32_; fz.gru_rh = hx2_; fz.gru_ld = Lhx; fz.add2 = hoist_ ? gz_[half] + rows * 256 : nullptr;
            // (upd8_: the fp8 copy of the [h | motion] slice starts 384 halfs after the pixel's first channel - conv()'s a8_rel - and the
            // epilogues store the new copies of r * h / h at byte 768 of the row)
            const int a8 = upd8_ && hoist_ ? 384 : 0;
            if ((r = conv(hx_, gc, Lhx, ND, h8_, w8_, kh, kw, 1, zr_[half], zrb_, 256, ACT_GRU_ZR, 0, hoist_ ? gz_[half] : nullptr, &fz, 0, a8, upd8_ ? 768 : 0))) return r;
            ConvFuse fq; fq.gru_h = h32_; fq.gru_z = zrb_; fq.add2 = hoist_ ? gq_[half] + rows * Lhx : nullptr;
            if ((r = conv(hx2_, gc, Lhx, ND, h8_, w8_, kh, kw, 1, q_[half], hx_, Lhx, ACT_GRU_Q, 0, hoist_ ? gq_[half] : nullptr, &fq, 0, a8, upd8_ ? 768 : 0))) return r;
        }
        // FlowHead -> delta_flow (fp32), coords1 += delta (the h slice's fp8 copy sits 384 halfs after it)
        if ((r = conv(hx_, 128, Lhx, ND, h8_, w8_, 3, 3, 1, fh1_, fh_, 256, ACT_RELU, 0, nullptr, nullptr, 0, upd8_ ? 384 : 0))) return r;
        tic(F_CONV, 2.0 * rows * 2.0 * fh2_.Kreal, 0);
        r = launch_flow_head2(stream, fh_, fh2_.w, fh2_.bias, flow_, ND, h8_, w8_, fh2_.sw);
        if (timer.enabled && !r) timer.recs[open_.back()].name = fh2_.sw ? "flow_head2_kernel<true>" : "flow_head2_kernel<false>";
        toc();
        if (r) return r;
        if (debug && it == 0 && (r = snapshot("flow_it0", flow_, (size_t)ND * P_ * 2 * 4, Stage{nullptr, 0, 0, P_, 1, 2, 2, (int64_t)P_ * 2}))) return r;
    }
    stages_["flow_lo"] = Stage{flow_, 0, 0, P_, 1, 2, 2, (int64_t)P_ * 2};

    // ---- mask head (last iteration only) + convex upsample + unpad + encode ----
    if ((r = conv(hx_, 128, Lhx, ND, h8_, w8_, 3, 3, 1, mk0_, m0_, 256, ACT_RELU, 0, nullptr, nullptr, 0, upd8_ ? 384 : 0))) return r;
    {
        GemmArgs a;
        a.A = m0_; a.lda = 256; a.N = 576; a.M = (int)rows;
        set_weights(a, mk2_, false);
        a.out32 = mask_; a.ldo = 576; a.scale = 0.25f;
        tic(F_GEMM, 2.0 * rows * 576.0 * 256, 0, 1.0 + mk2_.sw);
        r = launch_gemm(stream, A_DENSE, EPI_F32, TILE_AUTO, a);
        if (timer.enabled && !r && !open_.empty()) timer.recs[open_.back()].name = pb_gemm_last_kernel();
        toc();
        if (r) return r;
    }
    float *up = flow_out ? flow_out : up_;
    tic(F_PP, 0, (double)ND * sh_ * sw_ * 8);
    r = launch_upsample(stream, flow_, mask_, ND, h8_, w8_, padl_, padt_, sh_, sw_, up, maxd_);
    toc();
    if (r) return r;
    tic(F_PP, 0, (double)ND * sh_ * sw_ * 11);
    r = launch_flow_encode(stream, up, ND, sh_, sw_, maxd_, rgb_out, maxdisp);
    toc();
    if (r || !mask_out) return r;
    tic(F_PP, 0, (double)ND * sh_ * sw_ * 17);
    r = launch_fwdbwd_mask(stream, up, F - 1, sh_, sw_, alpha1, alpha2, mask_out);
    toc();
    return r;
}

int RaftEngine::snapshot(const char *name, const void *src, size_t bytes, const Stage &as) {
    void *&p = snaps_[name];
    if (p) PB_HIP(hipFree(p));
    p = nullptr;
    PB_HIP(hipMalloc(&p, bytes));
    PB_HIP(hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToDevice, stream));
    Stage st = as;
    st.ptr = p;
    stages_[name] = st;
    return 0;
}

int64_t RaftEngine::get_stage(const char *name, float *out, int64_t cap, int64_t shape[4]) {
    auto it = stages_.find(name);
    PB_CHECK(it != stages_.end(), PB_ERR_ARG, "unknown stage '%s'", name);
    const Stage &s = it->second;
    PB_HIP(hipStreamSynchronize(stream));
    if (s.kind == 0) {
        const int64_t total = (int64_t)last_nd_ * s.c * s.w;
        shape[0] = last_nd_; shape[1] = s.c; shape[2] = s.w; shape[3] = 1;
        PB_CHECK(total <= cap, PB_ERR_ARG, "stage buffer too small");
        PB_HIP(hipMemcpy(out, s.ptr, total * 4, hipMemcpyDeviceToHost));
        return total;
    }
    const int n = s.n > 0 ? (int)s.n : pF_;
    const int64_t total = (int64_t)n * s.c * s.h * s.w;
    shape[0] = n; shape[1] = s.c; shape[2] = s.h; shape[3] = s.w;
    PB_CHECK(total <= cap, PB_ERR_ARG, "stage buffer too small");
    float *tmp = nullptr;
    PB_HIP(hipMalloc((void **)&tmp, total * 4));
    int r = launch_nhwc_f16_to_nchw_f32(stream, (const f16 *)s.ptr, tmp, n, (int)s.c, (int)s.h, (int)s.w, (int)s.ld);
    if (r) return r;
    PB_HIP(hipStreamSynchronize(stream));
    PB_HIP(hipMemcpy(out, tmp, total * 4, hipMemcpyDeviceToHost));
    PB_HIP(hipFree(tmp));
    return total;
}
