// DepthEngine, metric mode: the ZoeDepth head over the Depth-Anything core (`bands/depth_anything.py --metric`).
// Reference: bands/depth_anything.py:106-119 (ToTensor, model, get_depth_from_prediction, Pillow resize),
// patchfusion/zoedepth/models/zoedepth/zoedepth_v1.py:139-215 (ZoeDepth.forward), base_models/depth_anything.py:
// 176-275 (DepthAnythingCore: 392 x 518 input, hooks out_conv / l4_rn / r4..r1), layers/{attractor,dist_layers,
// localbins_layers}.py.  Every 1x1 convolution of the head is a dense MFMA GEMM on the NHWC maps the DPT head left in
// HBM; bin centres, attractor points and the log-binomial stay fp32.
#include <math.h>

#include <algorithm>
#include <vector>

#include "engine.h"

namespace {
enum { F_GEMM = 0, F_ELT = 4, F_PP = 5 };

double pil_bicubic(double x) {           // libImaging/Resample.c bicubic_filter, a = -0.5
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

int pil_ksize(int in, int out) {
    double fs = (double)in / out;
    if (fs < 1.0) fs = 1.0;
    return (int)ceil(2.0 * fs) * 2 + 1;
}

// precompute_coeffs (Resample.c) for the whole axis: bounds {first index, count} and ksize coefficients per output
void pil_coeffs(int in, int out, std::vector<int> &bounds, std::vector<double> &kk) {
    const double scale = (double)in / out;
    const double fs = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * fs, ss = 1.0 / fs;
    const int ks = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out * 2, 0);
    kk.assign((size_t)out * ks, 0.0);
    for (int xx = 0; xx < out; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in) xmax = in;
        xmax -= xmin;
        double *k = &kk[(size_t)xx * ks];
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            const double w = pil_bicubic((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        bounds[(size_t)xx * 2] = xmin;
        bounds[(size_t)xx * 2 + 1] = xmax;
    }
}
}  // namespace

int DepthEngine::load_metric() {
    PB_CHECK(cfg_.features == 256 && cfg_.embed_dim == 1024, PB_ERR_ARG, "the metric head is built on the ViT-L core only");
    auto conv1 = [&](const std::string &name, int co, int ci, int kpad, PackedW &out) -> int {
        const pb_tensor *tw = find(name + ".weight"), *tb = find(name + ".bias");
        PB_CHECK(tw && tb && tw->shape[0] == co && tw->shape[1] == ci, PB_ERR_ARG, "missing or mis-shaped weight '%s' [%d, %d]",
                 name.c_str(), co, ci);
        int r = pack((const float *)tw->data, co, ci, kpad, out, (const float *)tb->data, 1, 0, vit_sw_);
        return r;
    };
    auto mlp = [&](const std::string &name, int cin, int kin, int mid, int kmid, int cout, Mlp2 &m) -> int {
        int r = conv1(name + ".0", mid, cin, kin, m.a);
        if (r) return r;
        return conv1(name + ".2", cout, mid, kmid, m.b);
    };
    int r;
    if ((r = conv1("conv2", 256, 256, 256, zconv2_))) return r;
    if ((r = mlp("seed_bin_regressor._net", 256, 256, 256, 256, 64, zseed_))) return r;
    if ((r = mlp("seed_projector._net", 256, 256, 128, 128, 128, zsproj_))) return r;
    const int na[4] = {16, 8, 4, 1};
    for (int i = 0; i < 4; ++i) {
        if ((r = mlp("projectors." + std::to_string(i) + "._net", 256, 256, 128, 128, 128, zproj_[i]))) return r;
        if ((r = mlp("attractors." + std::to_string(i) + "._net", 128, 128, 128, 128, na[i], zattr_[i]))) return r;
        zattr_[i].b.N = (int)round_up(na[i], 8);                 // GEMM writes 8-column groups; the pad rows are zero
    }
    if ((r = mlp("conditional_log_binomial.mlp", 161, 192, 80, 128, 4, zclb_))) return r;
    zclb_.b.N = 8;
    return 0;
}

int DepthEngine::plan_metric(int B, int H, int W) {
    const int64_t rows_f = round_up((int64_t)B * nh_ * nw_, 256);
    const int64_t rows_1 = round_up((int64_t)B * 4 * lh_[0] * lw_[0], 256);          // r1 resolution (224 x 296)
    act32_ = (f16 *)carve((size_t)rows_f * 32 * 2);
    zx0_ = (f16 *)carve((size_t)round_up((int64_t)B * lh_[3] * lw_[3], 256) * 256 * 2);
    zh_ = (f16 *)carve((size_t)rows_1 * 256 * 2);
    zemb_[0] = (f16 *)carve((size_t)rows_1 * 128 * 2); zemb_[1] = (f16 *)carve((size_t)rows_1 * 128 * 2);
    zxa_ = (f16 *)carve((size_t)rows_1 * 128 * 2);
    zA_ = (float *)carve((size_t)rows_1 * 16 * 4);
    zbins_[0] = (float *)carve((size_t)rows_1 * 64 * 4); zbins_[1] = (float *)carve((size_t)rows_1 * 64 * 4);
    zcat_ = (f16 *)carve((size_t)rows_f * 192 * 2);
    zmid_ = (f16 *)carve((size_t)rows_f * 128 * 2);
    zpt_ = (float *)carve((size_t)rows_f * 8 * 4);
    md_ = (float *)carve((size_t)B * nh_ * nw_ * 4);
    ptmp_ = (float *)carve((size_t)B * nh_ * W * 4);
    pxks_ = pil_ksize(nw_, W); pyks_ = pil_ksize(nh_, H);
    pxb_ = (int *)carve((size_t)W * 8); pyb_ = (int *)carve((size_t)H * 8);
    pxk_ = (double *)carve((size_t)W * pxks_ * 8); pyk_ = (double *)carve((size_t)H * pyks_ * 8);
    return 0;
}

int DepthEngine::metric_tables(int H, int W) {
    std::vector<int> xb, yb;
    std::vector<double> xk, yk;
    pil_coeffs(nw_, W, xb, xk);
    pil_coeffs(nh_, H, yb, yk);
    PB_HIP(hipMemcpyAsync(pxb_, xb.data(), xb.size() * 4, hipMemcpyHostToDevice, stream));
    PB_HIP(hipMemcpyAsync(pyb_, yb.data(), yb.size() * 4, hipMemcpyHostToDevice, stream));
    PB_HIP(hipMemcpyAsync(pxk_, xk.data(), xk.size() * 8, hipMemcpyHostToDevice, stream));
    PB_HIP(hipMemcpyAsync(pyk_, yk.data(), yk.size() * 8, hipMemcpyHostToDevice, stream));
    PB_HIP(hipStreamSynchronize(stream));           // the host vectors go out of scope
    return 0;
}

int DepthEngine::metric_head(int n) {
    int r;
    auto dense16 = [&](const f16 *A, int lda, int64_t M, const PackedW &w, f16 *out, int ldo, int act) -> int {
        GemmArgs a;
        a.A = A; a.lda = lda; a.M = (int)M; a.out = out; a.ldo = ldo; a.act = act;
        return gemm(A_DENSE, EPI_STD, a, w);
    };
    auto dense32 = [&](const f16 *A, int lda, int64_t M, const PackedW &w, float *out, int ldo) -> int {
        GemmArgs a;
        a.A = A; a.lda = lda; a.M = (int)M; a.out32 = out; a.ldo = ldo; a.scale = 1.f;
        return gemm(A_DENSE, EPI_F32, a, w);
    };
    auto softplus = [&](float *x, int64_t rows, int cols, int ld) -> int {
        tic(F_ELT, 0, (double)rows * cols * 8);
        int rr = launch_softplus(stream, x, rows, cols, ld);
        toc();
        return rr;
    };
    // the two bin buffers ping-pong, so intermediate levels are only kept (copied) for parity dumps
    auto keep_bins = [&](const std::string &name, const float *src, int hh, int ww) -> int {
        if (!debug) return 0;
        const size_t bytes = (size_t)n * hh * ww * 64 * 4;
        float *&p = snaps_["zoe." + name];
        if (p) PB_HIP(hipFree(p));
        p = nullptr;
        PB_HIP(hipMalloc((void **)&p, bytes));
        PB_HIP(hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToDevice, stream));
        stages_[name] = Stage{p, 0, last_n_, (int64_t)hh * ww, 1, 64, 64, (int64_t)hh * ww * 64};
        return 0;
    };
    const int Fp = 256 * hs_;            // row stride of the DPT head's maps ([hi | lo] per pixel in split-fp16 mode; hi is read)
    // ---- bottleneck, seed bins, seed embedding (zoedepth_v1.py:170-182) ----
    int h = lh_[3], w = lw_[3];
    int64_t rows = (int64_t)n * h * w;
    if ((r = dense16(rnraw_[3], Fp, rows, zconv2_, zx0_, 256, ACT_NONE))) return r;
    if ((r = dense16(zx0_, 256, rows, zseed_.a, zh_, 256, ACT_RELU))) return r;
    if ((r = dense32(zh_, 256, rows, zseed_.b, zbins_[0], 64))) return r;
    if ((r = softplus(zbins_[0], rows, 64, 64))) return r;
    if ((r = dense16(zx0_, 256, rows, zsproj_.a, zh_, 128, ACT_RELU))) return r;
    if ((r = dense16(zh_, 128, rows, zsproj_.b, zemb_[0], 128, ACT_NONE))) return r;
    if ((r = keep_bins("seed_bins", zbins_[0], h, w))) return r;
    int cb = 0, ce = 0;
    const int na[4] = {16, 8, 4, 1};
    // ---- projector + attractor per decoder level r4, r3, r2, r1 (zoedepth_v1.py:185-190) ----
    for (int i = 0; i < 4; ++i) {
        const f16 *xb = path_[3 - i];
        const int H = i == 3 ? 2 * lh_[0] : lh_[2 - i], W = i == 3 ? 2 * lw_[0] : lw_[2 - i];
        const int64_t R = (int64_t)n * H * W;
        if ((r = dense16(xb, Fp, R, zproj_[i].a, zh_, 128, ACT_RELU))) return r;
        if ((r = dense16(zh_, 128, R, zproj_[i].b, zemb_[ce ^ 1], 128, ACT_NONE))) return r;
        tic(F_ELT, 0, (double)R * 128 * 6);
        r = launch_bilerp_add(stream, zemb_[ce ^ 1], zemb_[ce], zxa_, n, h, w, H, W, 128, 128, 128, 128);
        toc();
        if (r) return r;
        if ((r = dense16(zxa_, 128, R, zattr_[i].a, zh_, 128, ACT_RELU))) return r;
        const int ldA = zattr_[i].b.N;
        if ((r = dense32(zh_, 128, R, zattr_[i].b, zA_, ldA))) return r;
        if ((r = softplus(zA_, R, na[i], ldA))) return r;
        tic(F_ELT, 0, (double)R * 64 * 8);
        r = launch_attractor(stream, zA_, ldA, na[i], zbins_[cb], h, w, zbins_[cb ^ 1], n, H, W, 300.f);   // inv_attractor's defaults apply
        toc();
        if (r) return r;
        cb ^= 1; ce ^= 1; h = H; w = W;
        if ((r = keep_bins("bins" + std::to_string(i), zbins_[cb], h, w))) return r;
    }
    // ---- conditional log-binomial over [out_conv activation | rel depth | embedding] (zoedepth_v1.py:192-205) ----
    const int64_t RF = (int64_t)n * nh_ * nw_;
    tic(F_ELT, 0, (double)RF * (192 * 2 + 128 * 8));
    r = launch_zoe_cat(stream, act32_, 32, netd_, zemb_[ce], 128, h, w, zcat_, n, nh_, nw_);
    toc();
    if (r) return r;
    if ((r = dense16(zcat_, 192, RF, zclb_.a, zmid_, 128, ACT_GELU))) return r;
    if ((r = dense32(zmid_, 128, RF, zclb_.b, zpt_, 8))) return r;
    tic(F_ELT, 0, (double)RF * (8 * 4 + 64 * 16));
    r = launch_logbinom_depth(stream, zpt_, 8, zbins_[cb], h, w, md_, n, nh_, nw_, 0.0212f, 50.f);
    toc();
    if (r) return r;
    stages_["metric_net"] = Stage{md_, 2, last_n_, 1, nh_, nw_, 0, 0};
    return 0;
}
