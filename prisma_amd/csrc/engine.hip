// DepthEngine: the depth_anything band on one MI355X.
// Reference call stack being replaced: bands/depth_anything.py:100-143 (infer) ->
// bands/d_anything/dpt.py:155-166 (DPT_DINOv2.forward) -> vision_transformer.py:297-321 /
// dinov2/layers/* (ViT) -> dpt.py:103-136 (DPTHead.forward) -> bands/depth_anything.py:132 and
// :215-221 (resize to the frame, min/max, heat encode).
#include "engine.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

namespace {
// families follow the kernel symbols rocprofv3 reports: gemm8_kernel<0,0,0> (fp16-out GEMMs: fc1, DPT 1x1 / convT),
// <0,1,0> (residual-accumulating proj / fc2), <0,2,0> (qkv), <1,0,0> (implicit-GEMM convs), attnq_kernel, ...
const char *kFam[] = {"gemm_f16", "conv_igemm_f16", "attention", "layernorm", "elementwise", "prepost", "gemm_f16_resid",
                      "gemm_f16_qkv"};
enum { F_GEMM = 0, F_CONV = 1, F_ATTN = 2, F_LN = 3, F_ELT = 4, F_PP = 5, F_GEMM_RESID = 6, F_GEMM_QKV = 7, F_COUNT = 8 };
inline int cp64(int c) { return (int)round_up(c, 64); }
}  // namespace

hipEvent_t KernelTimer::get() {
    if (used == pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        pool.push_back(e);
    }
    return pool[used++];
}
KernelTimer::~KernelTimer() {
    for (auto e : pool) hipEventDestroy(e);
}

void DepthEngine::tic(int fam, double flops, double bytes, double passes) {
    if (!timer.enabled) return;
    KernelTimer::Rec r{fam, timer.get(), timer.get(), flops, bytes, flops * passes, nullptr};
    hipEventRecord(r.a, stream);
    timer.recs.push_back(r);
}
void DepthEngine::toc() {
    if (!timer.enabled) return;
    hipEventRecord(timer.recs.back().b, stream);
}

int KernelTimer::collect(const char *const *fam_names, int nfam, pb_kernel_stat *out, int cap) {
    std::vector<pb_kernel_stat> acc;
    auto slot = [&](const char *name) -> pb_kernel_stat & {
        for (auto &a : acc)
            if (a.name == name || !strcmp(a.name, name)) return a;
        acc.push_back(pb_kernel_stat{name, 0, 0, 0, 0, 0});
        return acc.back();
    };
    for (auto &r : recs) {
        float ms = 0;
        hipEventElapsedTime(&ms, r.a, r.b);
        pb_kernel_stat &a = slot(r.name && r.name[0] ? r.name : fam_names[r.fam < nfam ? r.fam : 0]);
        a.ms += ms; a.flops += r.flops; a.exec_flops += r.exec; a.bytes += r.bytes; a.launches++;
    }
    int n = 0;
    for (auto &a : acc)
        if (n < cap) out[n++] = a;
    return n;
}

int DepthEngine::stats(pb_kernel_stat *out, int cap) {
    if (hipStreamSynchronize(stream) != hipSuccess) return -2;
    return timer.collect(kFam, F_COUNT, out, cap);
}

DepthEngine::DepthEngine(int dev, const pb_depth_cfg &cfg) : device(dev), cfg_(cfg) {}

void pb_rows_slice_major(f16 *rows, int64_t nrows, int64_t rowlen, int taps) {
    const int64_t S = rowlen / (64 * (int64_t)taps);
    std::vector<f16> row((size_t)rowlen);
    for (int64_t n = 0; n < nrows; ++n) {
        f16 *r = rows + n * rowlen;
        for (int t = 0; t < taps; ++t)
            for (int64_t sidx = 0; sidx < S; ++sidx) std::copy_n(r + ((int64_t)t * S + sidx) * 64, 64, row.data() + (sidx * taps + t) * 64);
        std::copy_n(row.data(), (size_t)rowlen, r);
    }
}

unsigned char pb_f32_to_e4m3(float x) {
    const unsigned char s = x < 0.f ? 0x80 : 0x00;
    const float a = fabsf(x);
    if (!(a == a)) return 0x7F;
    if (a >= 448.f) return s | 0x7E;
    int e;
    const float m = frexpf(a, &e);                  // a = m 2^e, m in [0.5, 1)
    int E = e - 1;                                  // a = (2 m) 2^E
    if (a == 0.f || E < -6) {                       // subnormal range: multiples of 2^-9
        const int q = (int)nearbyintf(a * 512.f);
        return s | (unsigned char)(q >= 8 ? 0x08 : q);
    }
    int q = (int)nearbyintf((2.f * m - 1.f) * 8.f);
    if (q == 8) { q = 0; ++E; }
    unsigned char code = (unsigned char)(((E + 7) << 3) | q);
    if (E > 8 || code > 0x7E) code = 0x7E;
    return s | code;
}

int DepthEngine::pack_mx(const float *src, int N, int K, PackedW &out, const float *bias) {
    PB_CHECK(K % 128 == 0, PB_ERR_ARG, "pack_mx: K %d", K);
    const int64_t Np = round_up(N, 256), ld = K + K / 2;           // halfs per row: K fp16 + K fp8 bytes
    float mlo = 0.f;
    for (int64_t i = 0; i < (int64_t)N * K; ++i) {
        const float v = src[i];
        mlo = fmaxf(mlo, fabsf(v - (float)(f16)v));
    }
    int pw = 0;
    if (mlo > 0.f) { int e; frexpf(mlo, &e); pw = 8 - e; }          // max |w_lo| 2^pw in [128, 256)
    std::vector<f16> h((size_t)Np * ld, (f16)0.f);
    for (int n = 0; n < N; ++n) {
        f16 *d = h.data() + (int64_t)n * ld;
        unsigned char *d8 = (unsigned char *)(d + K);
        for (int k = 0; k < K; ++k) {
            const float v = src[(int64_t)n * K + k];
            const f16 hi = (f16)v;
            d[k] = hi;
            d8[k] = pb_f32_to_e4m3(ldexpf(v - (float)hi, pw));
        }
    }
    void *p = nullptr;
    PB_HIP(hipMalloc(&p, h.size() * 2));
    owned_.push_back(p);
    PB_HIP(hipMemcpy(p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    out.w = (f16 *)p; out.N = N; out.K = (int)ld; out.Kreal = K; out.sa = 0; out.sw = 0; out.Cseg = K;
    out.nk16 = K / 64; out.mx_pw = pw;
    out.bias = nullptr;
    if (bias) return upload_f32(bias, N, &out.bias);
    return 0;
}

DepthEngine::~DepthEngine() {
    hipSetDevice(device);
    if (stream) hipStreamSynchronize(stream);
    for (auto p : owned_) hipFree(p);
    for (auto &kv : snaps_) hipFree(kv.second);
    if (arena_) hipFree(arena_);
    if (stream) hipStreamDestroy(stream);
}

int DepthEngine::dev_alloc(void **p, size_t bytes) {
    PB_HIP(hipMalloc(p, bytes));
    return 0;
}

const pb_tensor *DepthEngine::find(const std::string &name) const {
    auto it = tmap_.find(name);
    return it == tmap_.end() ? nullptr : it->second;
}

int DepthEngine::upload_f32(const float *src, size_t n, float **dst) {
    void *p = nullptr;
    PB_HIP(hipMalloc(&p, std::max<size_t>(n * 4, 256)));
    owned_.push_back(p);
    PB_HIP(hipMemcpy(p, src, n * 4, hipMemcpyHostToDevice));
    *dst = (float *)p;
    return 0;
}

// src: host fp32 [N, K] row-major (already in GEMM order; K = taps x channels).  Pads rows to a multiple of 256 and every
// tap to Kpad / taps channels.  Split-fp16 (sa / sw): each tap becomes the segments [w_hi | w_hi | w_lo] (PackedW, gemm.h).
int DepthEngine::pack(const float *src, int N, int K, int Kpad, PackedW &out, const float *bias, int taps, int sa, int sw) {
    const int64_t Np = round_up(N, 256);
    const int segs = 1 + sa + sw, Cin = K / taps, Cp = Kpad / taps;
    PB_CHECK(K % taps == 0 && Kpad % taps == 0 && Cp >= Cin && (segs == 1 || Cp % 64 == 0), PB_ERR_ARG, "pack: K %d / Kpad %d / taps %d", K, Kpad, taps);
    // 3x3 convolutions: slice-major K order (gemm.h cTapInner) - the taps of a 64-channel slice back to back, so the 256-channel maps'
    // lines are re-used out of the XCD L2 instead of being fetched once per tap
    const int tapin = head_tapin_ && taps > 1 && Cp % 64 == 0 ? 1 : 0;
    out.tapin = tapin; out.taps = taps;
    if (head_mx_ && sa && sw) {        // mx3 layout: per tap [w_hi fp16 | w_lo e4m3 2^pw | w_hi e4m3 2^(pw - 12)]
        float mlo = 0.f, mhi = 0.f;
        for (int64_t i = 0; i < (int64_t)N * K; ++i) {
            const float v = src[i];
            mhi = fmaxf(mhi, fabsf(v));
            mlo = fmaxf(mlo, fabsf(v - (float)(f16)v));
        }
        int pw = 0, e = 0;
        if (mlo > 0.f) { frexpf(mlo, &e); pw = 8 - e; }
        if (mhi > 0.f) { frexpf(mhi, &e); pw = std::min(pw, 20 - e); }     // max |w_hi| 2^(pw - 12) < 256 too
        const int64_t Kt = (int64_t)taps * 2 * Cp;
        std::vector<f16> h((size_t)Np * Kt, (f16)0.f);
        for (int n = 0; n < N; ++n)
            for (int t = 0; t < taps; ++t) {
                const float *s = src + (int64_t)n * K + (int64_t)t * Cin;
                f16 *d = h.data() + (int64_t)n * Kt + (int64_t)t * 2 * Cp;
                unsigned char *d8 = (unsigned char *)(d + Cp);
                for (int k = 0; k < Cin; ++k) {
                    const f16 hi = (f16)s[k];
                    d[k] = hi;
                    d8[k] = pb_f32_to_e4m3(ldexpf(s[k] - (float)hi, pw));
                    d8[Cp + k] = pb_f32_to_e4m3(ldexpf((float)hi, pw - 12));
                }
            }
        if (tapin) pb_rows_slice_major(h.data(), Np, Kt, taps);
        void *p = nullptr;
        PB_HIP(hipMalloc(&p, h.size() * 2));
        owned_.push_back(p);
        PB_HIP(hipMemcpy(p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        out.w = (f16 *)p; out.N = N; out.K = (int)Kt; out.Kreal = K; out.sa = 1; out.sw = 1; out.Cseg = Cp; out.mx3 = 1; out.mx_pw = pw;
        out.nk16 = Cp / 64;
        out.bias = nullptr;
        if (bias) return upload_f32(bias, N, &out.bias);
        return 0;
    }
    const int64_t Kt = (int64_t)taps * segs * Cp;
    std::vector<f16> h((size_t)Np * Kt, (f16)0.f);
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < taps; ++t) {
            const float *s = src + (int64_t)n * K + (int64_t)t * Cin;
            f16 *d = h.data() + (int64_t)n * Kt + (int64_t)t * segs * Cp;
            for (int k = 0; k < Cin; ++k) {
                const f16 hi = (f16)s[k];
                d[k] = hi;
                if (sa) d[Cp + k] = hi;
                if (sw) d[(1 + sa) * Cp + k] = (f16)(s[k] - (float)hi);
            }
        }
    if (tapin) pb_rows_slice_major(h.data(), Np, Kt, taps);
    void *p = nullptr;
    PB_HIP(hipMalloc(&p, h.size() * 2));
    owned_.push_back(p);
    PB_HIP(hipMemcpy(p, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    out.w = (f16 *)p;
    out.N = N;
    out.K = (int)Kt;
    out.Kreal = K;
    out.sa = sa; out.sw = sw; out.Cseg = Cp;
    out.bias = nullptr;
    if (bias) return upload_f32(bias, N, &out.bias);
    return 0;
}

int DepthEngine::load(const pb_tensor *w, int n) {
    PB_HIP(hipSetDevice(device));
    PB_HIP(pb_create_stream(&stream, "PB_CU_MASK_DEPTH"));
    for (int i = 0; i < n; ++i) {
        PB_CHECK(w[i].dtype == PB_F32 && w[i].data && w[i].name, PB_ERR_ARG, "weight %d: only host float32 tensors", i);
        tmap_[w[i].name] = &w[i];
    }
    const int D = cfg_.embed_dim, Hd = 4 * D, F = cfg_.features;
    if (const char *e = getenv("PB_TAPIN")) head_tapin_ = e[0] != '0';
    PB_CHECK(cfg_.precision == PB_PREC_F16 || cfg_.precision == PB_PREC_SPLIT, PB_ERR_ARG, "precision %d unknown", cfg_.precision);
    if (cfg_.precision == PB_PREC_SPLIT) {
        vit_sw_ = 1; head_sa_ = 1; head_sw_ = 1;
        // experiment switches (tools/precision_budget.py rows): PB_SPLIT=<vit_w><head_a><head_w>, e.g. 110
        if (const char *e = getenv("PB_SPLIT")) {
            if (strlen(e) == 3) { vit_sw_ = e[0] == '1'; head_sa_ = e[1] == '1'; head_sw_ = e[2] == '1'; }
        }
        hs_ = head_sa_ ? 2 : 1;
        const char *mx = getenv("PB_MX");
        vit_mx_ = vit_sw_ && D % 128 == 0 && !(mx && mx[0] == '0');
        // Which ViT linears keep their weight residual (bit 0 qkv, 1 proj, 2 fc1, 3 fc2).  tools/precision_budget_vit_layers.py (ViT-L 360x640,
        // everything else as the split mode runs it; max / range, L2): all four 3.9e-4 / 2.2e-4; without fc1's 4.2e-4 / 2.6e-4, without
        // qkv's 4.8e-4 / 2.9e-4, without fc2's 5.3e-4 / 3.3e-4, without proj's 5.9e-4 / 3.8e-4, none 7.9e-4 / 5.0e-4: the two layers that
        // read LayerNorm outputs are the cheap ones to drop - 58 % of the ViT linears' FLOPs run one pass instead of 1.5, and the LayerNorm
        // kernels stop writing e4m3 copies - for ~5.1e-4 / 3.2e-4 in quadrature.  Default 10 = proj + fc2; PB_VIT_RES=15 is round 2's mode.
        vit_res_ = vit_mx_ ? pb_env_int("PB_VIT_RES", 10) & 15 : 15;
        head_mx_ = head_sa_ && head_sw_ && !(mx && (mx[0] == '0' || mx[1] == '0'));      // PB_MX=10: MX in the ViT only
    }
    PB_CHECK(D % 128 == 0 && D <= 1024 && D / cfg_.heads == 64, PB_ERR_ARG, "embed_dim %d / heads %d unsupported", D,
             cfg_.heads);
    PB_CHECK(F % 64 == 0 && cfg_.depth >= 4, PB_ERR_ARG, "features %d / depth %d unsupported", F, cfg_.depth);
    auto need = [&](const std::string &name, int64_t numel) -> const float * {
        const pb_tensor *t = find(name);
        if (!t) { pb_set_error("missing weight '%s'", name.c_str()); return nullptr; }
        int64_t e = 1;
        for (int i = 0; i < t->ndim; ++i) e *= t->shape[i];
        if (e != numel) { pb_set_error("weight '%s': %lld elements, expected %lld", name.c_str(), (long long)e, (long long)numel); return nullptr; }
        return (const float *)t->data;
    };
#define NEED(var, name, numel) const float *var = need(name, numel); if (!var) return PB_ERR_ARG
#define UP(dst, name, numel) { NEED(_p, name, numel); int _r = upload_f32(_p, numel, &dst); if (_r) return _r; }
    // zero page
    {
        void *z = nullptr;
        PB_HIP(hipMalloc(&z, 4096));
        PB_HIP(hipMemset(z, 0, 4096));
        owned_.push_back(z);
        zero_ = (f16 *)z;
    }
    if ((cfg_.max_batch <= 1 || pb_env_int("PB_SPLITK_ALWAYS", 0)) && pb_env_int("PB_SPLITK", 8) > 1) {      // split-K workspace (engine.h sk_ws_)
        void *w = nullptr;
        PB_HIP(hipMalloc(&w, (size_t)kSkFloats * 4));
        owned_.push_back(w);
        sk_ws_ = (float *)w;
    }
    const std::string P = "pretrained.";
    {   // patch embedding: conv weight [D,3,14,14] is already [N, K=588] in (c, py, px) order
        NEED(wt, P + "patch_embed.proj.weight", (int64_t)D * 588);
        NEED(bs, P + "patch_embed.proj.bias", D);
        int r = pack(wt, D, 588, 640, patch_, bs, 1, 0, vit_sw_);
        if (r) return r;
        UP(cls_, P + "cls_token", D);
        const int g = cfg_.pos_grid;
        NEED(pe, P + "pos_embed", (int64_t)(1 + g * g) * D);
        pos_host_.assign(pe, pe + (int64_t)(1 + g * g) * D);
    }
    blocks_.resize(cfg_.depth);
    for (int i = 0; i < cfg_.depth; ++i) {
        const std::string b = P + "blocks." + std::to_string(i) + ".";
        Block &B = blocks_[i];
        UP(B.ln1g, b + "norm1.weight", D); UP(B.ln1b, b + "norm1.bias", D);
        UP(B.ln2g, b + "norm2.weight", D); UP(B.ln2b, b + "norm2.bias", D);
        UP(B.ls1, b + "ls1.gamma", D);     UP(B.ls2, b + "ls2.gamma", D);
        int r;
        { NEED(wt, b + "attn.qkv.weight", (int64_t)3 * D * D); NEED(bs, b + "attn.qkv.bias", 3 * D);
          if (vit_mx_ && !(vit_res_ & 1)) r = pack(wt, 3 * D, D, D, B.qkv, bs, 1, 0, 0);       // single fp16 pass (vit_res_)
          else r = vit_mx_ ? pack_mx(wt, 3 * D, D, B.qkv, bs) : pack(wt, 3 * D, D, D, B.qkv, bs, 1, 0, vit_sw_);
          if (r) return r; }
        // LayerScale (layer_scale.py:27-28) is folded into the weights: x + g*(W y + b) = x + (g.W) y + g.b, so the GEMM
        // can accumulate straight onto the residual stream
        auto pack_scaled = [&](const float *wt, const float *bs, const float *g, int N, int K, PackedW &out, bool keep_res) -> int {
            std::vector<float> ws((size_t)N * K), bb(N);
            for (int n = 0; n < N; ++n) {
                for (int k = 0; k < K; ++k) ws[(size_t)n * K + k] = wt[(size_t)n * K + k] * g[n];
                bb[n] = bs[n] * g[n];
            }
            if (vit_mx_ && !keep_res) return pack(ws.data(), N, K, K, out, bb.data(), 1, 0, 0);
            return vit_mx_ ? pack_mx(ws.data(), N, K, out, bb.data()) : pack(ws.data(), N, K, K, out, bb.data(), 1, 0, vit_sw_);
        };
        { NEED(wt, b + "attn.proj.weight", (int64_t)D * D); NEED(bs, b + "attn.proj.bias", D); NEED(g1, b + "ls1.gamma", D);
          if ((r = pack_scaled(wt, bs, g1, D, D, B.proj, (vit_res_ & 2) != 0))) return r; }
        { NEED(wt, b + "mlp.fc1.weight", (int64_t)Hd * D); NEED(bs, b + "mlp.fc1.bias", Hd);
          if (vit_mx_ && !(vit_res_ & 4)) {
              r = pack(wt, Hd, D, D, B.fc1, bs, 1, 0, 0);
              // no fp8 tiles of its own, but fc2 wants the e4m3 copy of fc1's output: that epilogue lives in the MX build of the kernel,
              // which nk16 = K / 64 selects (every K tile is an fp16 tile; gemm.h nk16)
              if (!r && (vit_res_ & 8)) B.fc1.nk16 = D / 64;
          } else {
              r = vit_mx_ ? pack_mx(wt, Hd, D, B.fc1, bs) : pack(wt, Hd, D, D, B.fc1, bs, 1, 0, vit_sw_);
          }
          if (r) return r; }
        { NEED(wt, b + "mlp.fc2.weight", (int64_t)D * Hd); NEED(bs, b + "mlp.fc2.bias", D); NEED(g2, b + "ls2.gamma", D);
          if ((r = pack_scaled(wt, bs, g2, D, Hd, B.fc2, (vit_res_ & 8) != 0))) return r; }
    }
    UP(normg_, P + "norm.weight", D);
    UP(normb_, P + "norm.bias", D);

    // ---- DPT head ----
    const std::string Hh = "depth_head.";
    const int *oc = cfg_.out_channels;
    // conv weight [co, ci, kh, kw] -> [co, (ky*kw + kx) * cip + ci]
    auto pack_conv = [&](const std::string &name, int co, int ci, int ks, bool has_bias, PackedW &out) -> int {
        const float *wt = need(name + ".weight", (int64_t)co * ci * ks * ks);
        if (!wt) return PB_ERR_ARG;
        const float *bs = nullptr;
        if (has_bias && !(bs = need(name + ".bias", co))) return PB_ERR_ARG;
        const int cip = cp64(ci), K = ks * ks * cip;
        std::vector<float> g((size_t)co * K, 0.f);
        for (int o = 0; o < co; ++o)
            for (int c = 0; c < ci; ++c)
                for (int t = 0; t < ks * ks; ++t) g[(size_t)o * K + t * cip + c] = wt[((size_t)o * ci + c) * ks * ks + t];
        int r = pack(g.data(), co, K, K, out, bs, ks * ks, head_sa_, head_sw_);
        out.Kreal = ks * ks * ci;
        return r;
    };
    // ConvTranspose2d weight [ci, co, s, s] (kernel == stride) -> rows n = (dy*s + dx)*co + o, k = ci
    auto pack_convT = [&](const std::string &name, int c, int s, PackedW &out) -> int {
        const float *wt = need(name + ".weight", (int64_t)c * c * s * s);
        const float *bs = need(name + ".bias", c);
        if (!wt || !bs) return PB_ERR_ARG;
        const int cip = cp64(c);
        std::vector<float> g((size_t)s * s * c * cip, 0.f);
        for (int ci = 0; ci < c; ++ci)
            for (int o = 0; o < c; ++o)
                for (int t = 0; t < s * s; ++t) g[((size_t)t * c + o) * cip + ci] = wt[((size_t)ci * c + o) * s * s + t];
        int r = pack(g.data(), s * s * c, cip, cip, out, nullptr, 1, head_sa_, head_sw_);
        out.Kreal = c;
        if (r) return r;
        return upload_f32(bs, c, &out.bias);
    };
    int r;
    for (int i = 0; i < 4; ++i) {
        if ((r = pack_conv(Hh + "projects." + std::to_string(i), oc[i], D, 1, true, proj_[i]))) return r;
        if ((r = pack_conv(Hh + "scratch.layer" + std::to_string(i + 1) + "_rn", F, oc[i], 3, false, rn_[i]))) return r;
        const std::string rf = Hh + "scratch.refinenet" + std::to_string(i + 1) + ".";
        if ((r = pack_conv(rf + "out_conv", F, F, 1, true, outc_[i]))) return r;
        for (int u = 0; u < 2; ++u)
            for (int c = 0; c < 2; ++c)
                if ((r = pack_conv(rf + "resConfUnit" + std::to_string(u + 1) + ".conv" + std::to_string(c + 1), F, F, 3,
                                   true, rcu_[i][u][c])))
                    return r;
    }
    if ((r = pack_convT(Hh + "resize_layers.0", oc[0], 4, rs0_))) return r;
    if ((r = pack_convT(Hh + "resize_layers.1", oc[1], 2, rs1_))) return r;
    if ((r = pack_conv(Hh + "resize_layers.3", oc[3], oc[3], 3, true, rs3_))) return r;
    if ((r = pack_conv(Hh + "scratch.output_conv1", F / 2, F, 3, true, oc1_))) return r;
    if ((r = pack_conv(Hh + "scratch.output_conv2.0", 32, F / 2, 3, true, oc2_))) return r;
    head_tail_ = pb_env_int("PB_HEAD_TAIL", 1) && !cfg_.metric;
    if (head_tail_) {
        const int ci = F / 2, cip = cp64(ci);
        const float *wt = need(Hh + "scratch.output_conv2.0.weight", (int64_t)32 * ci * 9);
        if (!wt) return PB_ERR_ARG;
        std::vector<float> g((size_t)288 * cip, 0.f);
        for (int c = 0; c < 32; ++c)
            for (int k = 0; k < ci; ++k)
                for (int t = 0; t < 9; ++t) g[((size_t)t * 32 + c) * cip + k] = wt[((size_t)c * ci + k) * 9 + t];
        if ((r = pack(g.data(), 288, cip, cip, wz_, nullptr, 1, head_sa_, head_sw_))) return r;
        wz_.Kreal = ci;
    }
    UP(w2_, Hh + "scratch.output_conv2.2.weight", 32);
    { NEED(b2, Hh + "scratch.output_conv2.2.bias", 1); b2_ = b2[0]; }
#undef NEED
#undef UP
    if (cfg_.metric) {
        int rm = load_metric();
        if (rm) return rm;
    }
    tmap_.clear();            // host tensors are not referenced after pb_create returns
    PB_HIP(hipDeviceSynchronize());   // null-stream memsets / copies done before the ctx stream is used
    return 0;
}

// ------------------------------------------------------------------------------------------------
// planning
// ------------------------------------------------------------------------------------------------
void *DepthEngine::carve(size_t bytes) {
    const size_t off = arena_off_;
    arena_off_ += round_up((int64_t)bytes, 256);
    return planning_ ? nullptr : (void *)(arena_ + off);
}

namespace {
// OpenCV INTER_CUBIC taps (see oracle/depth_oracle.py cubic_taps): float32 coefficients, A = -0.75.
void cubic_taps(int src, int dst, std::vector<int> &idx, std::vector<float> &wt) {
    idx.resize((size_t)dst * 4);
    wt.resize((size_t)dst * 4);
    const double scale = 1.0 / ((double)dst / (double)src);
    const float A = -0.75f;
    for (int d = 0; d < dst; ++d) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        const int sx = (int)floorf(fx);
        fx -= (float)sx;
        float c[4];
        c[0] = ((A * (fx + 1.f) - 5.f * A) * (fx + 1.f) + 8.f * A) * (fx + 1.f) - 4.f * A;
        c[1] = ((A + 2.f) * fx - (A + 3.f)) * fx * fx + 1.f;
        c[2] = ((A + 2.f) * (1.f - fx) - (A + 3.f)) * (1.f - fx) * (1.f - fx) + 1.f;
        c[3] = 1.f - c[0] - c[1] - c[2];
        for (int t = 0; t < 4; ++t) {
            idx[(size_t)d * 4 + t] = std::min(std::max(sx - 1 + t, 0), src - 1);
            wt[(size_t)d * 4 + t] = c[t];
        }
    }
}

inline float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
inline float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// torch F.interpolate(mode="bilinear", align_corners=True) as a 4-tap table (two zero weights): the metric model's
// input resize (patchfusion/zoedepth/models/base_models/depth_anything.py:172-174)
void bilinear_ac_taps(int src, int dst, std::vector<int> &idx, std::vector<float> &wt) {
    idx.assign((size_t)dst * 4, 0);
    wt.assign((size_t)dst * 4, 0.f);
    const float scale = dst > 1 ? (float)(src - 1) / (float)(dst - 1) : 0.f;
    for (int d = 0; d < dst; ++d) {
        const float s = scale * (float)d;
        int i0 = (int)s;
        if (i0 > src - 1) i0 = src - 1;
        const int i1 = i0 + (i0 < src - 1 ? 1 : 0);
        const float l = s - (float)i0;
        idx[(size_t)d * 4 + 0] = i0; idx[(size_t)d * 4 + 1] = i1; idx[(size_t)d * 4 + 2] = i1; idx[(size_t)d * 4 + 3] = i1;
        wt[(size_t)d * 4 + 0] = 1.f - l; wt[(size_t)d * 4 + 1] = l;
    }
}

// torch F.interpolate(mode="bicubic", scale_factor=(sy, sx)) of the [g, g, D] position grid to [gh, gw, D]
// (vision_transformer.py:179-210; numeric semantics verified in SURVEY.md section 8 a-4).
void interp_pos(const std::vector<float> &pe, int g, int D, int gh, int gw, float offset, std::vector<float> &out) {
    out.assign((size_t)(1 + gh * gw) * D, 0.f);
    memcpy(out.data(), pe.data(), (size_t)D * 4);
    if (gh == g && gw == g) {
        memcpy(out.data(), pe.data(), out.size() * 4);
        return;
    }
    const float A = -0.75f;
    const float ry = (float)(1.0 / ((double)(gh + offset) / g)), rx = (float)(1.0 / ((double)(gw + offset) / g));
    for (int oy = 0; oy < gh; ++oy) {
        const float sy = ry * ((float)oy + 0.5f) - 0.5f;
        const int iy = (int)floorf(sy);
        const float ty = sy - (float)iy;
        const float cy[4] = {cc2(ty + 1.f, A), cc1(ty, A), cc1(1.f - ty, A), cc2(2.f - ty, A)};
        for (int ox = 0; ox < gw; ++ox) {
            const float sx = rx * ((float)ox + 0.5f) - 0.5f;
            const int ix = (int)floorf(sx);
            const float tx = sx - (float)ix;
            const float cx[4] = {cc2(tx + 1.f, A), cc1(tx, A), cc1(1.f - tx, A), cc2(2.f - tx, A)};
            float *o = out.data() + (size_t)(1 + oy * gw + ox) * D;
            for (int i = 0; i < 4; ++i) {
                const int yy = std::min(std::max(iy - 1 + i, 0), g - 1);
                for (int j = 0; j < 4; ++j) {
                    const int xx = std::min(std::max(ix - 1 + j, 0), g - 1);
                    const float wgt = cy[i] * cx[j];
                    const float *s = pe.data() + (size_t)(1 + yy * g + xx) * D;
                    for (int c = 0; c < D; ++c) o[c] += wgt * s[c];
                }
            }
        }
    }
}
}  // namespace

void pb_cubic_taps(int src, int dst, int *idx, float *wt) {
    std::vector<int> i;
    std::vector<float> w;
    cubic_taps(src, dst, i, w);
    memcpy(idx, i.data(), i.size() * 4);
    memcpy(wt, w.data(), w.size() * 4);
}

int pb_depth_net_size(int H, int W, int *net_h, int *net_w) {
    PB_CHECK(H > 0 && W > 0 && net_h && net_w, PB_ERR_ARG, "net_size: bad arguments");
    // bands/d_anything/util/transform.py:100-166, lower_bound / keep_aspect_ratio / multiple of 14 / target 518
    double sh = 518.0 / H, sw = 518.0 / W;
    if (sw > sh) sh = sw; else sw = sh;
    auto constrain = [](double x) {
        int y = (int)(nearbyint(x / 14.0) * 14.0);          // np.round = round-half-even = nearbyint
        if (y < 518) y = (int)(ceil(x / 14.0) * 14.0);
        return y;
    };
    *net_h = constrain(sh * H);
    *net_w = constrain(sw * W);
    return 0;
}

int DepthEngine::batch_cap(int H, int W) const {
    int nh = 0, nw = 0;
    if (pb_depth_net_size(H, W, &nh, &nw)) return 1;
    if (cfg_.metric) { nh = 392; nw = 518; }
    const int64_t Fp = cp64(cfg_.features), F2p = cp64(cfg_.features / 2);
    const int64_t per = std::max((int64_t)nh * nw * F2p, (int64_t)16 * (nh / 14) * (nw / 14) * 4 * Fp) * hs_;   // elements of the largest map of one frame
    return (int)std::max<int64_t>(1, ((1LL << 31) - 1) / per);
}

int DepthEngine::prepare(int B, int H, int W) {
    if (B <= pB_ && H == pH_ && W == pW_) return 0;
    B = std::max(B, (pH_ == H && pW_ == W) ? pB_ : 0);
    PB_HIP(hipStreamSynchronize(stream));
    pB_ = 0; pH_ = 0; pW_ = 0;          // (a failure below must not leave the old plan's key on a half-written plan: raft_engine.hip prepare)
    int r = pb_depth_net_size(H, W, &nh_, &nw_);
    if (r) return r;
    if (cfg_.metric) { nh_ = 392; nw_ = 518; }      // DepthAnythingCore.prep: img_size [392, 518], aspect ratio not kept
    gh_ = nh_ / 14; gw_ = nw_ / 14; P_ = gh_ * gw_; ntok_ = P_ + 1; ntp_ = (int)round_up(ntok_, 16);
    lh_[0] = 4 * gh_; lw_[0] = 4 * gw_; lh_[1] = 2 * gh_; lw_[1] = 2 * gw_; lh_[2] = gh_; lw_[2] = gw_;
    lh_[3] = (gh_ - 1) / 2 + 1; lw_[3] = (gw_ - 1) / 2 + 1;
    const int D = cfg_.embed_dim, F = cfg_.features, Fp = cp64(F), F2p = cp64(F / 2);
    // GEMM row indices and per-tensor element offsets are 32-bit in the kernels.  The ViT's largest tensor (the fc1 output, 4 D x 1.5 halfs
    // per token) allows ~140 frames per launch; the DPT head's high-resolution split maps only batch_cap() of them (16 at 518 x 924): a call
    // runs the ViT on all B frames at once - its N = 1024 GEMMs have 2.4 rounds of tiles at 16 frames (3 rounds paid) but 4.8 at 32 -
    // and the head in chunks of HB frames (run_chunk)
    const int HB = std::min(B, batch_cap(H, W));
    hB_ = HB;
    PB_CHECK((int64_t)HB * nh_ * nw_ * F2p * hs_ < (1LL << 31) && (int64_t)HB * 4 * lh_[0] * lw_[0] * Fp * hs_ < (1LL << 31) &&
                 (int64_t)B * ntp_ * 4 * D * 3 / 2 < (1LL << 31), PB_ERR_ARG,
             "batch %d too large for %dx%d frames (32-bit tensor offsets); lower max_batch", B, H, W);
    const int64_t rows = round_up((int64_t)B * ntp_, 256);
    const size_t slack = 32768;
    for (int pass = 0; pass < 2; ++pass) {
        planning_ = pass == 0;
        arena_off_ = 0;
        xi_ = (int *)carve((size_t)nw_ * 16); xw_ = (float *)carve((size_t)nw_ * 16);
        yi_ = (int *)carve((size_t)nh_ * 16); yw_ = (float *)carve((size_t)nh_ * 16);
        pos_ = (float *)carve((size_t)ntok_ * D * 4);
        patchA_ = (f16 *)carve((size_t)round_up((int64_t)B * P_, 256) * 640 * 2);
        X_ = (float *)carve((size_t)rows * D * 4);
        const size_t mxs = vit_mx_ ? 3 : 2;          // bytes per element of a GEMM-input row: fp16 (+ an fp8 copy, gemm.h nk16)
        Y_ = (f16 *)carve((size_t)rows * D * mxs);
        const size_t qkb = (size_t)B * cfg_.heads * ntp_ * 64 * 2 + slack;
        Q_ = (f16 *)carve(qkb); K_ = (f16 *)carve(qkb); Vt_ = (f16 *)carve(qkb);
        AO_ = (f16 *)carve((size_t)rows * D * mxs);
        Hd_ = (f16 *)carve((size_t)rows * 4 * D * mxs);
        // split-fp16 head (hs_ = 2): every head map holds [hi | lo] per pixel / token
        for (int i = 0; i < 4; ++i) feat_[i] = (f16 *)carve((size_t)round_up((int64_t)B * P_, 256) * D * 2 * hs_);
        for (int i = 0; i < 4; ++i) {
            const size_t pix = (size_t)round_up((int64_t)HB * lh_[i] * lw_[i], 256) * hs_;
            const int ocp = cp64(cfg_.out_channels[i]);
            pj_[i] = (f16 *)carve((size_t)round_up((int64_t)HB * P_, 256) * ocp * 2 * hs_);
            lay_[i] = i == 2 ? pj_[2] : (f16 *)carve(pix * ocp * 2);
            rnraw_[i] = (f16 *)carve(pix * Fp * 2); rnrelu_[i] = (f16 *)carve(pix * Fp * 2);
            tmp_[i] = (f16 *)carve(pix * Fp * 2);
            sraw_[i] = (f16 *)carve(pix * Fp * 2); srelu_[i] = (f16 *)carve(pix * Fp * 2);
            yb_[i] = (f16 *)carve(pix * Fp * 2); ocb_[i] = (f16 *)carve(pix * Fp * 2);
            const int th = i == 0 ? 2 * lh_[0] : lh_[i - 1], tw = i == 0 ? 2 * lw_[0] : lw_[i - 1];
            path_[i] = (f16 *)carve((size_t)round_up((int64_t)HB * th * tw, 256) * Fp * 2 * hs_);
        }
        o1_ = (f16 *)carve((size_t)round_up((int64_t)HB * 4 * lh_[0] * lw_[0], 256) * F2p * 2 * hs_);
        if (head_tail_) z_ = (f16 *)carve((size_t)round_up((int64_t)HB * 4 * lh_[0] * lw_[0], 256) * cp64(288) * 2 * hs_);
        else up_ = (f16 *)carve((size_t)round_up((int64_t)HB * nh_ * nw_, 256) * F2p * 2 * hs_);
        netd_ = (float *)carve((size_t)HB * nh_ * nw_ * 4);
        full_ = (float *)carve((size_t)HB * H * W * 4);
        mm_ = (unsigned *)carve((size_t)HB * 8);
        if (cfg_.metric && (r = plan_metric(HB, H, W))) return r;
        if (pass == 0) {
            if (arena_off_ > arena_bytes_) {
                if (arena_) PB_HIP(hipFree(arena_));
                arena_ = nullptr;
                arena_bytes_ = 0;
                hipError_t e = hipMalloc((void **)&arena_, arena_off_);
                PB_CHECK(e == hipSuccess, PB_ERR_MEMORY, "arena of %zu bytes: %s", arena_off_, hipGetErrorString(e));
                arena_bytes_ = arena_off_;
            }
            // stale bit patterns must never be read as fp16 NaN/Inf by masked attention / padded tiles
            PB_HIP(hipMemsetAsync(arena_, 0, arena_bytes_, stream));
        }
    }
    std::vector<int> xi, yi;
    std::vector<float> xw, yw, pos;
    if (cfg_.metric) {
        bilinear_ac_taps(W, nw_, xi, xw);
        bilinear_ac_taps(H, nh_, yi, yw);
        if ((r = metric_tables(H, W))) return r;
    } else {
        cubic_taps(W, nw_, xi, xw);
        cubic_taps(H, nh_, yi, yw);
    }
    interp_pos(pos_host_, cfg_.pos_grid, D, gh_, gw_, 0.1f, pos);
    PB_HIP(hipMemcpyAsync(xi_, xi.data(), xi.size() * 4, hipMemcpyHostToDevice, stream));
    PB_HIP(hipMemcpyAsync(xw_, xw.data(), xw.size() * 4, hipMemcpyHostToDevice, stream));
    PB_HIP(hipMemcpyAsync(yi_, yi.data(), yi.size() * 4, hipMemcpyHostToDevice, stream));
    PB_HIP(hipMemcpyAsync(yw_, yw.data(), yw.size() * 4, hipMemcpyHostToDevice, stream));
    PB_HIP(hipMemcpyAsync(pos_, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, stream));
    PB_HIP(hipStreamSynchronize(stream));
    pB_ = B; pH_ = H; pW_ = W;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------------
int DepthEngine::gemm(int amode, int epi, GemmArgs &a, const PackedW &w, int tile) {
    a.W = w.w;
    a.K = w.K;
    if (w.mx3) {                 // [hi | hi8 | lo8] maps x [w_hi | w_lo8 | w_hi8] weights: fp16 tiles then fp8 tiles, per tap (gemm.h)
        a.nk16 = w.nk16; a.mx_scale_a = 127 - kLo8Pa; a.mx_scale_b = 127 - w.mx_pw;
        if (amode == A_CONV) {
            PB_CHECK(a.cC == w.Cseg, PB_ERR_STATE, "mx conv: %d channels per part, weights packed for %d", a.cC, w.Cseg);
            if (!a.cLd) a.cLd = 2 * a.cC;
            a.cC = 2 * w.Cseg;
            a.mx_period = 2 * w.Cseg / 64;
            if (w.tapin) { a.mx_period = 0; a.nk16 = w.nk16 * w.taps; }       // every fp16 slice of all taps, then the fp8 slices
        }
    } else if (w.sa || w.sw) {          // split-fp16 segments along K (gemm.h): the callers pass cC / lda of ONE part
        if (amode == A_CONV) {
            PB_CHECK(a.cC == w.Cseg, PB_ERR_STATE, "split conv: %d channels per part, weights packed for %d", a.cC, w.Cseg);
            if (!a.cLd) a.cLd = (1 + w.sa) * a.cC;
            a.cC = (1 + w.sa + w.sw) * w.Cseg;
            a.kwrap = w.sw ? (1 + w.sa) * w.Cseg : 0;
            a.kshift = -a.kwrap;
        } else {
            a.kwrap = w.sw ? (1 + w.sa) * w.Cseg / 64 : 0;
        }
    }
    if (amode == A_CONV && w.tapin) { a.cTapInner = 1; a.cKH = w.taps / a.cKW; }
    if (w.nk16 && !w.mx3) { a.nk16 = w.nk16; a.mx_scale_a = 127 - kMxPa; a.mx_scale_b = 127 - w.mx_pw; }
    if (a.lo_off && head_mx_) { a.lo8 = 1; a.lo8_pa = kLo8Pa; }
    if (!a.N) a.N = w.N;
    if (!a.bias) a.bias = w.bias;
    a.zero = zero_;
    const double flops = 2.0 * a.M * (double)a.N * w.Kreal;
    // algorithmic bytes: the A operand once (a convolution reads its input MAP once - not the im2col matrix: VERDICT r3 item 9), the
    // weights once, the output once (the fused head writes one float per pixel)
    const int taps = w.taps > 0 ? w.taps : 1;
    const double a_elems = amode == A_CONV ? (double)(a.M / ((int64_t)a.cOH * a.cOW)) * a.cH * a.cW * (w.Kreal / taps) : (double)a.M * w.Kreal;
    const double bytes = 2.0 * (a_elems + (double)a.N * w.Kreal) + (epi == EPI_HEAD ? 4.0 * a.M : 2.0 * (double)a.M * a.N);
    tic(amode == A_CONV ? F_CONV : (epi == EPI_RESID ? F_GEMM_RESID : (epi == EPI_QKV ? F_GEMM_QKV : F_GEMM)), flops, bytes, w.mx3 ? 2.0 : 1.0 + w.sa + w.sw + (w.nk16 && w.K > w.Kreal ? 0.5 : 0.0));
    if (tile == TILE_AUTO) tile = amode == A_CONV ? conv_tile : gemm_tile;
    if (sk_ws_) { a.sk_ws = sk_ws_; a.sk_cap = kSkFloats; }
    int r = launch_gemm(stream, amode, epi, tile, a);
    if (timer.enabled && !r) timer.recs.back().name = pb_gemm_last_kernel();
    toc();
    return r;
}

int DepthEngine::conv3(const f16 *in, int inC, int n, int H, int W, const PackedW &w, f16 *out, f16 *out2,
                       const f16 *add1, const f16 *add2, int act, int stride, int outC) {
    GemmArgs a;
    a.A = in;
    a.cH = H; a.cW = W; a.cC = inC; a.cKW = 3; a.cStride = stride; a.cPad = 1;
    a.cOH = (H + 2 - 3) / stride + 1; a.cOW = (W + 2 - 3) / stride + 1;
    a.M = n * a.cOH * a.cOW;
    a.out = out; a.out2 = out2; a.add1 = add1; a.add2 = add2; a.act = act; a.ldo = hs_ * outC;
    a.lo_off = head_sa_ ? outC : 0;
    return gemm(A_CONV, EPI_STD, a, w, TILE_AUTO);
}

void DepthEngine::snapshot(const std::string &name) {
    if (!debug) return;
    const size_t bytes = (size_t)pB_ * ntp_ * cfg_.embed_dim * 4;
    float *&p = snaps_[name];
    if (!p) hipMalloc((void **)&p, bytes);
    hipMemcpyAsync(p, X_, bytes, hipMemcpyDeviceToDevice, stream);
    stages_[name] = Stage{p, 0, last_n_, ntok_, 1, cfg_.embed_dim, cfg_.embed_dim, (int64_t)ntp_ * cfg_.embed_dim};
}

int DepthEngine::vit(int n) {
    const int D = cfg_.embed_dim, M = n * ntp_;
    int r;
    {   // patch embedding GEMM straight into the fp32 residual stream, + bias + interpolated pos-embed
        GemmArgs a;
        a.A = patchA_; a.lda = 640; a.M = n * P_;
        a.resid = X_; a.ldr = D; a.pos = pos_; a.ppi = P_; a.ntp = ntp_; a.D = D;
        if ((r = gemm(A_DENSE, EPI_PATCH, a, patch_, TILE_128))) return r;
        tic(F_ELT, 0, 0);
        r = launch_cls_rows(stream, X_, cls_, pos_, n, ntp_, D);
        toc();
        if (r) return r;
    }
    snapshot("tokens");
    const double ln_bytes = (double)n * ntok_ * D * 6.0;
    // MX mode: rows of Y_ / AO_ / Hd_ are [fp16 (K) | fp8 (K bytes)], i.e. 1.5 K halfs apart; the fp8 copy is stored x 2^kMxPa.  A buffer
    // whose consumers run without a weight residual (vit_res_) is plain fp16: Y_ feeds qkv (after norm1) and fc1 (after norm2)
    const bool y8 = vit_mx_ && (vit_res_ & 5) != 0, ao8 = vit_mx_ && (vit_res_ & 2), hd8 = vit_mx_ && (vit_res_ & 8);
    PB_CHECK(!vit_mx_ || (vit_res_ & 5) == 0 || (vit_res_ & 5) == 5, PB_ERR_ARG, "PB_VIT_RES: qkv and fc1 share their input buffer - keep or drop both residuals");
    const int ldy = vit_mx_ ? D + D / 2 : D, o8 = vit_mx_ ? 2 * D : 0;
    const int ldY = y8 ? ldy : D, o8Y = y8 ? o8 : 0, ldA = ao8 ? ldy : D, o8A = ao8 ? o8 : 0, ldH = hd8 ? 4 * ldy : 4 * D;
    const float o8s = (float)(1 << kMxPa);
    for (int i = 0; i < cfg_.depth; ++i) {
        const Block &b = blocks_[i];
        tic(F_LN, 0, ln_bytes);
        r = launch_layernorm(stream, X_, b.ln1g, b.ln1b, Y_, n, ntp_, ntok_, D, 1e-6f, 0, ldY, 0, o8Y, o8s);
        toc();
        if (r) return r;
        {
            GemmArgs a;
            a.A = Y_; a.lda = ldY; a.M = M;
            a.q = Q_; a.k = K_; a.vt = Vt_; a.ntp = ntp_; a.heads = cfg_.heads; a.D = D; a.qscale = PB_QSCALE;
            if ((r = gemm(A_DENSE, EPI_QKV, a, b.qkv))) return r;
        }
        tic(F_ATTN, 4.0 * n * cfg_.heads * (double)ntok_ * ntok_ * 64.0, (double)n * ntok_ * D * 2.0 * 4.0);
        r = launch_attention(stream, Q_, K_, Vt_, AO_, n, cfg_.heads, ntp_, ntok_, ldA, 0, o8A, o8s);
        toc();
        if (r) return r;
        {
            GemmArgs a;
            a.A = AO_; a.lda = ldA; a.M = M; a.resid = X_; a.ldr = D; a.gamma = b.ls1;
            if ((r = gemm(A_DENSE, EPI_RESID, a, b.proj))) return r;
        }
        tic(F_LN, 0, ln_bytes);
        r = launch_layernorm(stream, X_, b.ln2g, b.ln2b, Y_, n, ntp_, ntok_, D, 1e-6f, 0, ldY, 0, o8Y, o8s);
        toc();
        if (r) return r;
        {
            GemmArgs a;
            a.A = Y_; a.lda = ldY; a.M = M; a.out = Hd_; a.ldo = ldH; a.act = ACT_GELU;
            if (hd8) { a.o8_off = 4 * D * 2; a.o8_scale = o8s; }
            if ((r = gemm(A_DENSE, EPI_STD, a, b.fc1))) return r;
        }
        {
            GemmArgs a;
            a.A = Hd_; a.lda = ldH; a.M = M; a.resid = X_; a.ldr = D; a.gamma = b.ls2;
            if ((r = gemm(A_DENSE, EPI_RESID, a, b.fc2))) return r;
        }
        if (debug) snapshot("block" + std::to_string(i));
        const int tap = i - (cfg_.depth - 4);
        if (tap >= 0) {
            tic(F_LN, 0, ln_bytes);
            r = launch_layernorm(stream, X_, normg_, normb_, feat_[tap], n, ntp_, ntok_, D, 1e-6f, 1, hs_ * D, head_sa_ ? D : 0, 0, 16.f,
                                 head_mx_ ? kLo8Pa : -1);
            toc();
            if (r) return r;
            stages_["feat" + std::to_string(tap)] = Stage{feat_[tap], 3, last_n_, P_, 1, D, hs_ * D, (int64_t)P_ * D};
        }
    }
    return 0;
}

// frames [f0, f0 + n) of the ViT batch (the four normalised taps feat_[i] hold all of it)
int DepthEngine::head(int f0, int n) {
    const int D = cfg_.embed_dim, F = cfg_.features, Fp = cp64(F), F2 = F / 2, F2p = cp64(F2);
    const int *oc = cfg_.out_channels;
    int r;
    const int hs = hs_;                                  // 2: maps are [hi | lo] per pixel (split fp16), lo at + padded channels
    auto lo = [&](int cp) { return head_sa_ ? cp : 0; };
    auto nhwc = [&](const std::string &name, const f16 *p, int c, int h, int w, int ld, bool split = true) {
        stages_[name] = Stage{p, 1, last_n_, c, h, w, (split ? hs : 1) * ld, 0};
    };
    auto bil = [&](const f16 *x, f16 *y, int h, int w, int oh, int ow, int c, int ld) -> int {
        tic(F_ELT, 0, (double)n * ((double)h * w + (double)oh * ow) * c * 2.0 * hs);
        int rr = launch_bilinear_nhwc(stream, x, y, n, h, w, oh, ow, c, hs * ld, 1, lo(ld), head_mx_ ? kLo8Pa : -1);
        toc();
        return rr;
    };
    // reassemble: 1x1 projection of each tap, then x4 / x2 transposed convs, identity, 3x3 stride 2
    for (int i = 0; i < 4; ++i) {
        GemmArgs a;
        a.A = feat_[i] + (int64_t)f0 * P_ * hs * D; a.lda = hs * D; a.M = n * P_; a.out = pj_[i]; a.ldo = hs * cp64(oc[i]); a.lo_off = lo(cp64(oc[i]));
        if ((r = gemm(A_DENSE, EPI_STD, a, proj_[i]))) return r;
    }
    for (int i = 0; i < 2; ++i) {
        const int s = i == 0 ? 4 : 2;
        GemmArgs a;
        a.A = pj_[i]; a.lda = hs * cp64(oc[i]); a.M = n * P_;
        a.out = lay_[i]; a.ldo = hs * cp64(oc[i]); a.lo_off = lo(cp64(oc[i])); a.ps_s = s; a.ps_h = gh_; a.ps_w = gw_; a.ps_co = oc[i];
        if ((r = gemm(A_DENSE, EPI_PIXSHUF, a, i == 0 ? rs0_ : rs1_))) return r;
    }
    if ((r = conv3(pj_[3], cp64(oc[3]), n, gh_, gw_, rs3_, lay_[3], nullptr, nullptr, nullptr, ACT_NONE, 2, cp64(oc[3]))))
        return r;
    for (int i = 0; i < 4; ++i) {
        if ((r = conv3(lay_[i], cp64(oc[i]), n, lh_[i], lw_[i], rn_[i], rnraw_[i], rnrelu_[i], nullptr, nullptr, ACT_NONE,
                       1, Fp)))
            return r;
        nhwc("layer" + std::to_string(i + 1) + "_rn", rnraw_[i], F, lh_[i], lw_[i], Fp);
    }
    // refinenet4 .. refinenet1 (level 3 = coarsest)
    for (int lv = 3; lv >= 0; --lv) {
        const int h = lh_[lv], w = lw_[lv];
        const f16 *sr = rnraw_[lv], *sl = rnrelu_[lv];
        if (lv < 3) {
            // resConfUnit1 on the lateral, summed with the upsampled path from the coarser level
            if ((r = conv3(rnrelu_[lv], Fp, n, h, w, rcu_[lv][0][0], tmp_[lv], nullptr, nullptr, nullptr, ACT_RELU, 1, Fp))) return r;
            if ((r = conv3(tmp_[lv], Fp, n, h, w, rcu_[lv][0][1], sraw_[lv], srelu_[lv], rnraw_[lv], path_[lv + 1], ACT_NONE, 1, Fp))) return r;
            sr = sraw_[lv]; sl = srelu_[lv];
        }
        if ((r = conv3(sl, Fp, n, h, w, rcu_[lv][1][0], tmp_[lv], nullptr, nullptr, nullptr, ACT_RELU, 1, Fp))) return r;
        if ((r = conv3(tmp_[lv], Fp, n, h, w, rcu_[lv][1][1], yb_[lv], nullptr, sr, nullptr, ACT_NONE, 1, Fp))) return r;
        {   // out_conv (1x1) commutes with the bilinear resize (both linear, taps sum to 1): run it at low res
            GemmArgs a;
            a.A = yb_[lv]; a.lda = hs * Fp; a.M = n * h * w; a.out = ocb_[lv]; a.ldo = hs * Fp; a.lo_off = lo(Fp);
            if ((r = gemm(A_DENSE, EPI_STD, a, outc_[lv]))) return r;
        }
        const int th = lv == 0 ? 2 * h : lh_[lv - 1], tw = lv == 0 ? 2 * w : lw_[lv - 1];
        if ((r = bil(ocb_[lv], path_[lv], h, w, th, tw, F, Fp))) return r;
        nhwc("path" + std::to_string(lv + 1), path_[lv], F, th, tw, Fp);
    }
    const int h1 = 2 * lh_[0], w1 = 2 * lw_[0];
    if ((r = conv3(path_[0], Fp, n, h1, w1, oc1_, o1_, nullptr, nullptr, nullptr, ACT_NONE, 1, F2p))) return r;
    nhwc("output_conv1", o1_, F2, h1, w1, F2p);
    if (head_tail_) {   // output_conv2's nine tap products at the low resolution, then resize + sum + ReLU + 1 x 1 + ReLU in one pass (engine.h wz_)
        const int Zp = cp64(288);
        GemmArgs a;
        a.A = o1_; a.lda = hs * F2p; a.M = n * h1 * w1; a.out = z_; a.ldo = hs * Zp; a.lo_off = lo(Zp); a.N = 288;
        // (K = F / 2 is two to four K tiles: the launch is its epilogue - 1.5 ms per 16 frames on the 128 x 128 tile, 1.75 on the ping-pong kernel)
        if ((r = gemm(A_DENSE, EPI_STD, a, wz_))) return r;
        tic(F_ELT, 0, (double)n * ((double)h1 * w1 * 288 * (head_sa_ ? 3.0 : 2.0) + (double)nh_ * nw_ * 4.0));
        r = launch_dpt_tail(stream, z_, n, h1, w1, hs * Zp, lo(Zp), head_mx_ ? kLo8Pa : -1, oc2_.bias, w2_, b2_, netd_, nh_, nw_);
        toc();
        if (r) return r;
        stages_["net_depth"] = Stage{netd_, 2, last_n_, 1, nh_, nw_, 0, 0};
        return 0;
    }
    if ((r = bil(o1_, up_, h1, w1, nh_, nw_, F2, F2p))) return r;
    if (cfg_.metric) {   // the metric head needs the 32-channel activation ("out_conv" hook): 3x3 + ReLU, then the 1x1 + ReLU
        GemmArgs a;
        a.A = up_;
        a.cH = nh_; a.cW = nw_; a.cC = F2p; a.cKW = 3; a.cStride = 1; a.cPad = 1; a.cOH = nh_; a.cOW = nw_;
        a.M = n * nh_ * nw_; a.N = 32; a.out = act32_; a.ldo = 32; a.act = ACT_RELU;
        if ((r = gemm(A_CONV, EPI_STD, a, oc2_))) return r;
        tic(F_ELT, 0, (double)a.M * 68);
        r = launch_dot32_relu(stream, act32_, 32, w2_, b2_, netd_, a.M);
        toc();
        if (r) return r;
        nhwc("output_conv2_0", act32_, 32, nh_, nw_, 32, false);
    } else {   // output_conv2: 3x3 -> ReLU -> 1x1 -> ReLU fused in one implicit-GEMM launch
        GemmArgs a;
        a.A = up_;
        a.cH = nh_; a.cW = nw_; a.cC = F2p; a.cKW = 3; a.cStride = 1; a.cPad = 1; a.cOH = nh_; a.cOW = nw_;
        a.M = n * nh_ * nw_; a.N = 32;
        a.w2 = w2_; a.b2 = b2_; a.depth = netd_;
        if ((r = gemm(A_CONV, EPI_HEAD, a, oc2_))) return r;
    }
    stages_["net_depth"] = Stage{netd_, 2, last_n_, 1, nh_, nw_, 0, 0};
    return 0;
}

int DepthEngine::run_chunk(const uint8_t *frames, int n, float *depth_out, uint8_t *rgb_out, float *mn, float *mx,
                           int flip) {
    int r;
    stages_.clear();
    last_n_ = n;
    tic(F_PP, 0, (double)n * ((double)pH_ * pW_ * 3 + (double)nh_ * nw_ * 3 * 2));
    r = launch_preprocess(stream, frames, n, pH_, pW_, nh_, nw_, xi_, xw_, yi_, yw_, patchA_, 640, nullptr);
    toc();
    if (r) return r;
    if ((r = vit(n))) return r;
    // the DPT head and the tail in balanced chunks of at most hB_ frames (32-bit offsets of the high-resolution split maps)
    const int nch = (n + hB_ - 1) / hB_, step = (n + nch - 1) / nch;
    const int64_t px = (int64_t)pH_ * pW_;
    for (int f0 = 0; f0 < n; f0 += step) {
        const int c = std::min(step, n - f0);
        last_n_ = c;                                     // the head's stages hold the last head chunk
        if ((r = head(f0, c))) return r;
        if ((r = tail(c, depth_out ? depth_out + f0 * px : nullptr, rgb_out ? rgb_out + f0 * px * 3 : nullptr, mn ? mn + f0 : nullptr,
                      mx ? mx + f0 : nullptr, flip)))
            return r;
    }
    return 0;
}

// net depth of the n frames the head just produced -> band resize, per-frame min / max, heat encode (reference :132-133, :215-221)
int DepthEngine::tail(int n, float *depth_out, uint8_t *rgb_out, float *mn, float *mx, int flip) {
    int r;
    // dpt.py:163-164 (bilinear to the same size with align_corners=True, ReLU) is the identity on netd_.
    float *full = depth_out ? depth_out : full_;
    if (cfg_.metric) {      // ZoeDepth head -> Pillow resize to the frame -> per-frame min / max -> heat encode
        if ((r = metric_head(n))) return r;
        tic(F_PP, 0, (double)n * ((double)nh_ * nw_ * 4 + (double)pH_ * pW_ * 8));
        r = launch_pil_resize(stream, md_, ptmp_, full, n, nh_, nw_, pH_, pW_, pxb_, pxk_, pxks_, pyb_, pyk_, pyks_);
        if (!r) r = launch_init_minmax(stream, mm_, n);
        if (!r) r = launch_minmax_only(stream, full, n, (int64_t)pH_ * pW_, mm_);
        toc();
        if (r) return r;
        tic(F_PP, 0, (double)n * (double)pH_ * pW_ * 7);
        r = launch_heat_encode(stream, full, n, pH_, pW_, mm_, flip, rgb_out, mn, mx);
        toc();
        return r;
    }
    tic(F_PP, 0, (double)n * ((double)nh_ * nw_ * 4 + (double)pH_ * pW_ * 4));
    if ((r = launch_init_minmax(stream, mm_, n))) return r;
    r = launch_depth_resize_minmax(stream, netd_, n, nh_, nw_, full, pH_, pW_, mm_);
    toc();
    if (r) return r;
    tic(F_PP, 0, (double)n * (double)pH_ * pW_ * 7);
    r = launch_heat_encode(stream, full, n, pH_, pW_, mm_, flip, rgb_out, mn, mx);
    toc();
    return r;
}

int DepthEngine::infer(const uint8_t *frames, int n, int H, int W, float *depth_out, uint8_t *rgb_out, float *mn,
                       float *mx, int flip) {
    PB_CHECK(frames && n > 0 && H > 0 && W > 0, PB_ERR_ARG, "infer: bad arguments");
    PB_HIP(hipSetDevice(device));
    const int cap = std::max(1, cfg_.max_batch);         // frames per ViT launch; the head runs in chunks of batch_cap(H, W) inside (run_chunk)
    int r = prepare(std::min(n, cap), H, W);
    if (r) return r;
    timer.reset();
    const int step = (n + (n + cap - 1) / cap - 1) / ((n + cap - 1) / cap);      // balanced chunks of at most cap frames
    for (int i = 0; i < n; i += step) {
        const int c = std::min(step, n - i);
        const int64_t px = (int64_t)H * W;
        r = run_chunk(frames + i * px * 3, c, depth_out ? depth_out + i * px : nullptr,
                      rgb_out ? rgb_out + i * px * 3 : nullptr, mn ? mn + i : nullptr, mx ? mx + i : nullptr, flip);
        if (r) return r;
    }
    return 0;
}

int64_t DepthEngine::get_stage(const char *name, float *out, int64_t cap, int64_t shape[4]) {
    auto it = stages_.find(name);
    PB_CHECK(it != stages_.end(), PB_ERR_ARG, "unknown stage '%s'", name);
    const Stage &s = it->second;
    // frames the stage holds: the whole call for the ViT's stages, the LAST head chunk (frames n - s.n ... n - 1 of the call) for the
    // DPT head's when the call ran its head in more than one chunk - recorded when the stage was registered
    const int n = s.n > 0 ? (int)s.n : last_n_;
    PB_HIP(hipStreamSynchronize(stream));
    int64_t total = 0;
    if (s.kind == 0 || s.kind == 3) {
        total = (int64_t)n * s.c * s.w;                       // [n, rows = s.c, cols = s.w]
        shape[0] = n; shape[1] = s.c; shape[2] = s.w; shape[3] = 1;
        PB_CHECK(total <= cap, PB_ERR_ARG, "stage buffer too small (%lld > %lld)", (long long)total, (long long)cap);
        if (s.kind == 0) {
            for (int b = 0; b < n; ++b)
                PB_HIP(hipMemcpy(out + (int64_t)b * s.c * s.w, (const float *)s.ptr + b * s.bstride, (size_t)s.c * s.w * 4,
                                 hipMemcpyDeviceToHost));
        } else {
            float *tmp = nullptr;
            PB_HIP(hipMalloc((void **)&tmp, total * 4));
            int r = launch_f16_to_f32(stream, (const f16 *)s.ptr, tmp, (int64_t)n * s.c, (int)s.w, (int)s.ld);
            if (r) return r;
            PB_HIP(hipStreamSynchronize(stream));
            PB_HIP(hipMemcpy(out, tmp, total * 4, hipMemcpyDeviceToHost));
            PB_HIP(hipFree(tmp));
        }
    } else if (s.kind == 1) {
        total = (int64_t)n * s.c * s.h * s.w;
        shape[0] = n; shape[1] = s.c; shape[2] = s.h; shape[3] = s.w;
        PB_CHECK(total <= cap, PB_ERR_ARG, "stage buffer too small (%lld > %lld)", (long long)total, (long long)cap);
        float *tmp = nullptr;
        PB_HIP(hipMalloc((void **)&tmp, total * 4));
        int r = launch_nhwc_f16_to_nchw_f32(stream, (const f16 *)s.ptr, tmp, n, (int)s.c, (int)s.h, (int)s.w, (int)s.ld);
        if (r) return r;
        PB_HIP(hipStreamSynchronize(stream));
        PB_HIP(hipMemcpy(out, tmp, total * 4, hipMemcpyDeviceToHost));
        PB_HIP(hipFree(tmp));
    } else {
        total = (int64_t)n * s.h * s.w;
        shape[0] = n; shape[1] = s.h; shape[2] = s.w; shape[3] = 1;
        PB_CHECK(total <= cap, PB_ERR_ARG, "stage buffer too small (%lld > %lld)", (long long)total, (long long)cap);
        PB_HIP(hipMemcpy(out, s.ptr, total * 4, hipMemcpyDeviceToHost));
    }
    return total;
}
