#include "engine_base.h"
#include "conv_walk.h"      // cw3_tiles16 / cw3_tiles8: the packed-channel K axis the kernels walk

#include <math.h>

#include <algorithm>

namespace {
const char *kFam[] = {"gemm_f16", "conv_igemm_f16", "attention", "layernorm", "elementwise", "prepost", "conv_igemm_f16_tile128",
                      "conv_igemm_f16_tile256x64"};
inline int cp64(int c) { return (int)round_up(c, 64); }
}  // namespace

EngineBase::~EngineBase() {
    hipSetDevice(device);
    if (stream) hipStreamSynchronize(stream);
    for (auto p : owned_) hipFree(p);
    if (arena_) hipFree(arena_);
    if (stream) hipStreamDestroy(stream);
}

void EngineBase::tic(int fam, double flops, double bytes, double passes) {
    if (!timer.enabled) return;
    KernelTimer::Rec r{fam, timer.get(), timer.get(), flops, bytes, flops * passes, nullptr};
    hipEventRecord(r.a, cur_);
    timer.recs.push_back(r);
    open_.push_back(timer.recs.size() - 1);
}
void EngineBase::toc() {
    if (!timer.enabled) return;
    hipEventRecord(timer.recs[open_.back()].b, cur_);
    open_.pop_back();
}

int EngineBase::stats(pb_kernel_stat *out, int cap) {
    if (hipStreamSynchronize(stream) != hipSuccess) return -2;
    return timer.collect(kFam, F_COUNT, out, cap);
}

int EngineBase::begin_load(const pb_tensor *w, int n) {
    PB_HIP(hipSetDevice(device));
    PB_HIP(pb_create_stream(&stream, "PB_CU_MASK_FLOW"));        // (the convolutional bands: flow_raft, flow_gmflow, mask_mmdet)
    cur_ = stream;
    for (int i = 0; i < n; ++i) {
        PB_CHECK(w[i].data && w[i].name, PB_ERR_ARG, "weight %d: null", i);
        if (w[i].dtype == PB_F32) tmap_[w[i].name] = &w[i];      // integer buffers (num_batches_tracked) are not needed
    }
    void *z = nullptr;
    PB_HIP(hipMalloc(&z, 4096));
    PB_HIP(hipMemset(z, 0, 4096));
    owned_.push_back(z);
    zero_ = (f16 *)z;
    return 0;
}

const pb_tensor *EngineBase::find(const std::string &name) const {
    auto it = tmap_.find(name);
    return it == tmap_.end() ? nullptr : it->second;
}

int EngineBase::pack(const float *src, int N, int K, int Kpad, PackedW &out, const float *bias, int taps, int sa_req) {
    const int64_t Np = round_up(N, 256);
    // split_w_ (PB_PREC_SPLIT): every tap's channels are followed by the fp16 rounding residuals of the same weights, and the
    // GEMM reads the activations twice (gemm.h kwrap): a w_hi + a w_lo in one accumulator
    const int sw = split_w_ && Kpad % (64 * taps) == 0 && K % taps == 0 ? 1 : 0, sa = sw && sa_req ? 1 : 0, segs = 1 + sa + sw;
    const int Cin = sw ? K / taps : K, Cp = sw ? Kpad / taps : Kpad, tp = sw ? taps : 1;
    // slice-major K order (gemm.h cTapInner): rows are (taps x S) grids of 128-byte blocks in every layout below - transpose them
    const int tapin = pack_tapin_ && taps > 1 && Kpad % (64 * taps) == 0 ? 1 : 0;
    auto to_slice_major = [&](std::vector<f16> &h, int64_t rowlen) {
        if (tapin) pb_rows_slice_major(h.data(), Np, rowlen, taps);
    };
    out.tapin = tapin; out.taps = taps;
    if (!sa && sw && mx_ && pack_mx2_ && Cp % 128 == 0) {        // mx2 layout: per tap [w_hi fp16 | w_lo e4m3 2^pw]
        float mlo = 0.f;
        for (int64_t i = 0; i < (int64_t)N * K; ++i) mlo = fmaxf(mlo, fabsf(src[i] - (float)(f16)src[i]));
        int pw = 0, e = 0;
        if (mlo > 0.f) { frexpf(mlo, &e); pw = 8 - e; }
        const int64_t K2 = (int64_t)tp * (Cp + Cp / 2);
        std::vector<f16> h2((size_t)Np * K2, (f16)0.f);
        for (int n = 0; n < N; ++n)
            for (int t = 0; t < tp; ++t) {
                f16 *d = h2.data() + (size_t)n * K2 + (size_t)t * (Cp + Cp / 2);
                unsigned char *d8 = (unsigned char *)(d + Cp);
                for (int k = 0; k < Cin; ++k) {
                    const float v = src[(size_t)n * K + (size_t)t * Cin + k];
                    const f16 hi = (f16)v;
                    d[k] = hi;
                    d8[k] = pb_f32_to_e4m3(ldexpf(v - (float)hi, pw));
                }
            }
        to_slice_major(h2, K2);
        void *p2 = nullptr;
        PB_HIP(hipMalloc(&p2, h2.size() * 2));
        owned_.push_back(p2);
        PB_HIP(hipMemcpy(p2, h2.data(), h2.size() * 2, hipMemcpyHostToDevice));
        out.w = (f16 *)p2; out.N = N; out.K = (int)K2; out.Kreal = K; out.bias = nullptr;
        out.sa = 0; out.sw = 1; out.Cseg = Cp; out.mx2 = 1; out.mx_pw = pw; out.nk16 = Cp / 64;
    } else if (sa && mx_) {             // mx3 layout (engine.hip DepthEngine::pack has the same one)
        float mlo = 0.f, mhi = 0.f;
        for (int64_t i = 0; i < (int64_t)N * K; ++i) {
            const float v = src[i];
            mhi = fmaxf(mhi, fabsf(v));
            mlo = fmaxf(mlo, fabsf(v - (float)(f16)v));
        }
        int pw = 0, e = 0;
        if (mlo > 0.f) { frexpf(mlo, &e); pw = 8 - e; }
        if (mhi > 0.f) { frexpf(mhi, &e); pw = std::min(pw, 20 - e); }
        const int64_t K3 = (int64_t)tp * 2 * Cp;
        std::vector<f16> h3((size_t)Np * K3, (f16)0.f);
        for (int n = 0; n < N; ++n)
            for (int t = 0; t < tp; ++t) {
                f16 *d = h3.data() + (size_t)n * K3 + (size_t)t * 2 * Cp;
                unsigned char *d8 = (unsigned char *)(d + Cp);
                for (int k = 0; k < Cin; ++k) {
                    const float v = src[(size_t)n * K + (size_t)t * Cin + k];
                    const f16 hi = (f16)v;
                    d[k] = hi;
                    d8[k] = pb_f32_to_e4m3(ldexpf(v - (float)hi, pw));
                    d8[Cp + k] = pb_f32_to_e4m3(ldexpf((float)hi, pw - 12));
                }
            }
        to_slice_major(h3, K3);
        void *p3 = nullptr;
        PB_HIP(hipMalloc(&p3, h3.size() * 2));
        owned_.push_back(p3);
        PB_HIP(hipMemcpy(p3, h3.data(), h3.size() * 2, hipMemcpyHostToDevice));
        out.w = (f16 *)p3; out.N = N; out.K = (int)K3; out.Kreal = K; out.bias = nullptr;
        out.sa = 1; out.sw = 1; out.Cseg = Cp; out.mx3 = 1; out.mx_pw = pw; out.nk16 = Cp / 64;
    } else {
    const int64_t Kt = (int64_t)tp * segs * Cp;
    std::vector<f16> h((size_t)Np * Kt, (f16)0.f);
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < tp; ++t)
            for (int k = 0; k < Cin; ++k) {
                const float v = src[(size_t)n * K + (size_t)t * Cin + k];
                const f16 hi = (f16)v;
                f16 *d = h.data() + (size_t)n * Kt + (size_t)t * segs * Cp;
                d[k] = hi;
                if (sa) d[Cp + k] = hi;
                if (sw) d[(1 + sa) * Cp + k] = (f16)(v - (float)hi);
            }
    to_slice_major(h, Kt);
    void *p = nullptr;
    PB_HIP(hipMalloc(&p, h.size() * 2));
    owned_.push_back(p);
    PB_HIP(hipMemcpy(p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    out.w = (f16 *)p; out.N = N; out.K = (int)Kt; out.Kreal = K; out.bias = nullptr;
    out.sa = sa; out.sw = sw; out.Cseg = Cp;
    }
    if (bias) {
        void *b = nullptr;
        PB_HIP(hipMalloc(&b, std::max<size_t>((size_t)Np * 4, 256)));
        owned_.push_back(b);
        PB_HIP(hipMemset(b, 0, (size_t)Np * 4));
        PB_HIP(hipMemcpy(b, bias, (size_t)N * 4, hipMemcpyHostToDevice));
        out.bias = (float *)b;
    }
    return 0;
}

int EngineBase::fold_bn(const std::string &bn, int C, std::vector<float> &scale, std::vector<float> &shift) {
    const char *sfx[4] = {".weight", ".bias", ".running_mean", ".running_var"};
    const float *t[4];
    for (int i = 0; i < 4; ++i) {
        const pb_tensor *x = find(bn + sfx[i]);
        PB_CHECK(x && x->shape[0] == C, PB_ERR_ARG, "missing weight '%s%s' [%d]", bn.c_str(), sfx[i], C);
        t[i] = (const float *)x->data;
    }
    scale.resize(C); shift.resize(C);
    for (int c = 0; c < C; ++c) {
        const float s = t[0][c] / sqrtf(t[3][c] + 1e-5f);       // nn.BatchNorm2d eval, eps 1e-5
        scale[c] = s;
        shift[c] = t[1][c] - t[2][c] * s;
    }
    return 0;
}

int EngineBase::pack_conv(const std::string &name, bool has_bias, const float *scale, const float *shift, PackedW &out, int sa, int ci_pad) {
    const pb_tensor *t = find(name + ".weight");
    PB_CHECK(t && t->ndim == 4, PB_ERR_ARG, "missing conv '%s'", name.c_str());
    const float *b = nullptr;
    if (has_bias) {
        const pb_tensor *tb = find(name + ".bias");
        PB_CHECK(tb, PB_ERR_ARG, "missing bias of '%s'", name.c_str());
        b = (const float *)tb->data;
    }
    const int co = (int)t->shape[0], ci = (int)t->shape[1], kh = (int)t->shape[2], kw = (int)t->shape[3];
    const float *w = (const float *)t->data;
    const int cip = ci_pad ? ci_pad : cp64(ci), K = kh * kw * cip;
    PB_CHECK(cip >= ci && cip % 64 == 0, PB_ERR_ARG, "pack_conv '%s': %d input channels padded to %d", name.c_str(), ci, cip);
    std::vector<float> g((size_t)co * K, 0.f), bb(co);
    for (int o = 0; o < co; ++o) {
        const float s = scale ? scale[o] : 1.f;
        for (int c = 0; c < ci; ++c)
            for (int tp = 0; tp < kh * kw; ++tp) g[(size_t)o * K + tp * cip + c] = w[((size_t)o * ci + c) * kh * kw + tp] * s;
        bb[o] = (b ? b[o] : 0.f) * s + (shift ? shift[o] : 0.f);
    }
    int r = pack(g.data(), co, K, K, out, bb.data(), kh * kw, sa);
    out.Kreal = kh * kw * ci;
    if (r) return r;
    // RAFT's encoder stage 2: 96 -> 96 channels carried as 128.  The per-tap layout walks 9 x (2 + 2) K tiles of which a quarter multiply padding
    // channels; the packed-channel copy (conv_walk.h conv_cw3_word: 14 + 14 tiles) is what the 128 x 96 tile reads (PB_CW3=0: not built)
    static const int cw3 = pb_env_int("PB_CW3", 1);
    if (cw3 && out.mx3 && !out.tapin && kh * kw > 1 && ci % 16 == 0 && ci < cip && co > 64 && co <= 96) {
        const int taps = kh * kw, nk16 = cw3_tiles16(taps, ci), nk8 = cw3_tiles8(taps, ci), pw = out.mx_pw;
        const int64_t Kc = (int64_t)(nk16 + nk8) * 64, Np = round_up(co, 256);
        std::vector<f16> h((size_t)Np * Kc, (f16)0.f);
        for (int o = 0; o < co; ++o) {
            f16 *row = h.data() + (size_t)o * Kc;
            unsigned char *row8 = (unsigned char *)(row + (size_t)nk16 * 64);
            for (int tp = 0; tp < taps; ++tp)
                for (int c = 0; c < ci; ++c) {
                    const float v = g[(size_t)o * K + tp * cip + c];
                    const f16 hi = (f16)v;
                    row[tp * ci + c] = hi;                                                   // fp16 chunks: tap-major, ci / 8 per tap
                    row8[(tp * 2 + 0) * ci + c] = pb_f32_to_e4m3(ldexpf(v - (float)hi, pw));   // meets the map's hi8 chunks
                    row8[(tp * 2 + 1) * ci + c] = pb_f32_to_e4m3(ldexpf((float)hi, pw - 12));  // ... and its lo8 chunks
                }
        }
        void *pc = nullptr;
        PB_HIP(hipMalloc(&pc, h.size() * 2));
        owned_.push_back(pc);
        PB_HIP(hipMemcpy(pc, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        out.wcw = (f16 *)pc; out.Kcw = (int)Kc; out.nk16cw = nk16; out.cwC = ci;
    }
    return 0;
}

void EngineBase::set_weights(GemmArgs &a, const PackedW &w, bool is_conv) const {
    a.W = w.w; a.K = w.K; a.bias = w.bias; a.zero = zero_;
    const bool tapin = is_conv && w.tapin;
    if (tapin) { a.cTapInner = 1; a.cKH = w.taps / a.cKW; }
    if (w.mx2) {                 // [a16 | a8] maps x [w_hi | w_lo8] weights: per tap Cseg / 64 fp16 tiles then Cseg / 128 fp8 tiles
        a.nk16 = w.nk16; a.mx_scale_a = 127 - kMx2Pa; a.mx_scale_b = 127 - w.mx_pw;
        if (is_conv) {
            const int c = a.cC;                   // channels of the slice; the caller set cLd and (conv()) a8_rel via kshift
            a.cC = c + c / 2;
            a.mx_period = a.cC / 64;
            if (tapin) { a.mx_period = 0; a.nk16 = w.nk16 * w.taps; }      // slice-major: every fp16 slice of all taps, then the fp8 slices
        }
        return;
    }
    if (w.mx3) {                 // [hi | hi8 | lo8] maps: per tap fp16 tiles then fp8 tiles (gemm.h mx_period)
        a.nk16 = w.nk16; a.mx_scale_a = 127 - kLo8Pa; a.mx_scale_b = 127 - w.mx_pw;
        if (is_conv) {
            if (!a.cLd) a.cLd = 2 * a.cC;
            a.cC = 2 * w.Cseg;
            a.mx_period = 2 * w.Cseg / 64;
            if (tapin) { a.mx_period = 0; a.nk16 = w.nk16 * w.taps; }
            if (w.wcw) { a.Wcw = w.wcw; a.Kcw = w.Kcw; a.nk16cw = w.nk16cw; a.cwC = w.cwC; a.cwPad = w.Cseg; a.cwTaps = w.taps; }
        }
        return;
    }
    if (!w.sw) return;
    if (is_conv) {               // per tap [w_hi | w_hi (if sa) | w_lo]: the channel cursor wraps back onto the pixel's hi part
        if (!a.cLd) a.cLd = (1 + w.sa) * a.cC;
        a.kwrap = (1 + w.sa) * a.cC;
        a.kshift = -a.kwrap;
        a.cC = (2 + w.sa) * a.cC;
    } else {
        a.kwrap = (1 + w.sa) * w.Cseg / 64;
    }
}

void *EngineBase::carve(size_t bytes) {
    const size_t off = arena_off_;
    arena_off_ += round_up((int64_t)bytes, 256);
    return planning_ ? nullptr : (void *)(arena_ + off);
}

int EngineBase::commit_arena(const char *what) {
    if (arena_off_ > arena_bytes_) {
        if (arena_) PB_HIP(hipFree(arena_));
        arena_ = nullptr; arena_bytes_ = 0;
        hipError_t e = hipMalloc((void **)&arena_, arena_off_);
        PB_CHECK(e == hipSuccess, PB_ERR_MEMORY, "%s arena of %zu bytes: %s", what, arena_off_, hipGetErrorString(e));
        arena_bytes_ = arena_off_;
    }
    // stale bit patterns must never be read as fp16 NaN / Inf by padded tiles
    PB_HIP(hipMemsetAsync(arena_, 0, arena_bytes_, stream));
    return 0;
}

int EngineBase::conv(const f16 *in, int cC, int cLd, int n, int H, int W, int kh, int kw, int stride, const PackedW &w, f16 *out,
                     int ldo, int act, int pre_relu, const f16 *add1, const ConvFuse *fuse, int lo_off, int a8_rel, int o8_off) {
    GemmArgs a;
    a.A = in; a.N = w.N;
    a.cH = H; a.cW = W; a.cC = cC; a.cLd = cLd; a.cKW = kw; a.cStride = stride; a.cPad = kh / 2; a.cPadX = kw / 2;
    a.cOH = (H + 2 * (kh / 2) - kh) / stride + 1; a.cOW = (W + 2 * (kw / 2) - kw) / stride + 1;
    a.M = n * a.cOH * a.cOW;
    a.out = out; a.ldo = ldo; a.act = act; a.pre_relu = pre_relu; a.add1 = add1; a.lo_off = lo_off;
    if (lo_off && mx_) { a.lo8 = 1; a.lo8_pa = kLo8Pa; }
    if (fuse) { a.out2 = fuse->out2; a.gru_h = fuse->gru_h; a.gru_z = fuse->gru_z; a.gru_rh = fuse->gru_rh; a.gru_ld = fuse->gru_ld; a.add2 = fuse->add2; }
    if (w.mx2 && a8_rel && a8_rel != cC) { a.kwrap = cC; a.kshift = a8_rel - cC; }      // the fp8 copy of a channel slice is not adjacent
    if (o8_off) { a.o8_off = o8_off; a.o8_scale = (float)(1 << kMx2Pa); }
    PB_CHECK(!w.sw || w.Cseg == cC, PB_ERR_STATE, "split conv: %d channels, weights packed for %d", cC, w.Cseg);
    set_weights(a, w, true);
    PB_CHECK(w.K == kh * kw * a.cC, PB_ERR_STATE, "conv: packed K %d != %d*%d*%d", w.K, kh, kw, a.cC);
    // same predicate as launch_gemm's TILE_AUTO: the 256 x 256 ping-pong kernel needs N % 256 == 0 and >= 256 tiles
    const bool wide = conv_tile == TILE_256 || (conv_tile == TILE_AUTO && a.N % 256 == 0 && (int64_t)(a.M / 256) * (a.N / 256) >= 256);
    // algorithmic bytes: the input map once, the weights once, the output once (fp16)
    // one family per kernel symbol launch_gemm picks: 256 x 256 ping-pong, 256 x 64 (N <= 64), 128 x 128
    const double nr = w.Nreal ? w.Nreal : a.N;       // (padding rows of the weights are not the layer's work)
    tic(wide ? F_CONV : (conv_tile == TILE_AUTO && a.N <= 64 ? F_CONV64 : F_CONV128), 2.0 * a.M * nr * w.Kreal, 2.0 * ((double)n * H * W * cC + nr * w.Kreal + (double)a.M * nr), w.mx3 ? 2.0 : (w.mx2 ? 1.5 : 1.0 + w.sa + w.sw));
    // 64 -> 64 channels at half resolution: the halo-tiled direct kernel (halo_conv.hip) instead of the implicit GEMM
    int r = kh == 3 && kw == 3 && conv_tile == TILE_AUTO && conv3x3_c64_supported(a) ? launch_conv3x3_c64(cur_, a) : launch_gemm(cur_, A_CONV, EPI_STD, conv_tile, a);
    if (timer.enabled && !r) timer.recs[open_.back()].name = pb_gemm_last_kernel();
    toc();
    return r;
}

int EngineBase::dense(const f16 *A, int lda, int64_t M, const PackedW &w, f16 *out, int ldo, int act, const f16 *add1, int o8_off, int a_pa,
                      int lo_off) {
    GemmArgs a;
    a.A = A; a.lda = lda; a.N = w.N; a.M = (int)M;
    set_weights(a, w, false);
    a.out = out; a.ldo = ldo; a.act = act; a.add1 = add1; a.lo_off = lo_off;
    if (lo_off && mx_) { a.lo8 = 1; a.lo8_pa = kLo8Pa; }
    if (o8_off) { a.o8_off = o8_off; a.o8_scale = (float)(1 << kMx2Pa); }
    if (w.mx2 && a_pa >= 0) a.mx_scale_a = 127 - a_pa;
    tic(F_GEMM, 2.0 * M * (double)a.N * w.Kreal, 2.0 * ((double)M * w.Kreal + (double)a.N * w.Kreal + (double)M * a.N), w.mx3 ? 2.0 : (w.mx2 ? 1.5 : 1.0 + w.sa + w.sw));
    int r = launch_gemm(cur_, A_DENSE, EPI_STD, TILE_AUTO, a);
    if (timer.enabled && !r) timer.recs[open_.back()].name = pb_gemm_last_kernel();
    toc();
    return r;
}
