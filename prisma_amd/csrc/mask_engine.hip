// MaskEngine: the mask_mmdet band (SOLOv2 R-101 FPN) on one MI355X.
// Reference call stack being replaced: bands/mask_mmdet.py:131-154 (per-frame loop) -> mmdet/apis/inference.py:99-162
// (test pipeline + model(return_loss=False)) -> models/detectors/single_stage_instance_seg.py:150-250 ->
// models/backbones/resnet.py, models/necks/fpn.py, models/dense_heads/solov2_head.py (forward, get_results),
// core/post_processing/matrix_nms.py -> format_results -> bands/mask_mmdet.py:43-61,139-147 (accumulation).
//
// Schedule: frames go through backbone / neck / head in batches (the reference runs one frame at a time); every
// convolution is an MFMA GEMM launch (1x1 as dense GEMM, 3x3 as implicit GEMM) with eval-mode BatchNorm folded and
// ReLU / residual add in the epilogue; GroupNorm needs run-time statistics and stays a stats + apply pair.  The
// data-dependent part of get_results (which grid cells pass score_thr, the area filter, two sorts) is control
// logic on the host over a few KB of scores, exactly the decisions torch makes on the reference's host; every
// numeric step (sigmoid / points-NMS, dynamic convolution, mask areas and maskness, pairwise intersections,
// Matrix-NMS decay, the two bilinear resizes, thresholds and the accumulation) is a kernel.
#include "mask_engine.h"

#include <math.h>
#include <string.h>

#include <algorithm>

#include "raft_kernels.h"

namespace {
inline int cp64(int c) { return (int)round_up(c, 64); }

// OpenCV resize(INTER_LINEAR) tables for 8-bit images: {i0, i1, c0, c1} per destination index.  Columns zero the
// fraction at the borders, rows clamp the indices instead (imgproc resize.cpp, the linear branch).
void linear_taps_u8(int src, int dst, bool rows, std::vector<int> &t) {
    t.resize((size_t)dst * 4);
    const double scale = 1.0 / ((double)dst / (double)src);
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        int i0, i1;
        if (rows) {
            i0 = std::min(std::max(s, 0), src - 1);
            i1 = std::min(std::max(s + 1, 0), src - 1);
        } else {
            if (s < 0) { f = 0.f; s = 0; }
            if (s >= src - 1) { f = 0.f; s = src - 1; }
            i0 = s;
            i1 = std::min(s + 1, src - 1);
        }
        t[(size_t)d * 4 + 0] = i0;
        t[(size_t)d * 4 + 1] = i1;
        t[(size_t)d * 4 + 2] = (int)nearbyintf((1.f - f) * 2048.f);
        t[(size_t)d * 4 + 3] = (int)nearbyintf(f * 2048.f);
    }
}
}  // namespace

void MaskEngine::net_size(const pb_mask_cfg &cfg, int H, int W, int *nh, int *nw, int *Hp, int *Wp) {
    // mmcv.rescale_size((long, short)): factor = min(long / max(h, w), short / min(h, w)); int(x * factor + 0.5)
    const double f = std::min((double)cfg.scale_long / std::max(H, W), (double)cfg.scale_short / std::min(H, W));
    *nh = (int)(H * f + 0.5);
    *nw = (int)(W * f + 0.5);
    *Hp = (*nh + 31) / 32 * 32;
    *Wp = (*nw + 31) / 32 * 32;
}

MaskEngine::~MaskEngine() {
    hipSetDevice(device);
    if (stream) hipStreamSynchronize(stream);
    void *post[] = {pk_, bits_, plog_, pstat_, inter_, sig_, nmsf_, pidx_, nmsi_, use_, inst_, sdf_tab_, sdf_g_};
    for (auto p : post)
        if (p) hipFree(p);
    for (int l = 0; l < 5; ++l) {
        if (ls_[l]) { hipStreamSynchronize(ls_[l]); hipStreamDestroy(ls_[l]); }
        if (ev_join_[l]) hipEventDestroy(ev_join_[l]);
    }
    if (ev_fork_) hipEventDestroy(ev_fork_);
}

int MaskEngine::load_gn(const std::string &name, int C, GN &out) {
    auto ig = tmap_.find(name + ".weight"), ib = tmap_.find(name + ".bias");
    PB_CHECK(ig != tmap_.end() && ib != tmap_.end() && ig->second->shape[0] == C, PB_ERR_ARG, "missing group norm '%s' [%d]",
             name.c_str(), C);
    void *p = nullptr;
    PB_HIP(hipMalloc(&p, (size_t)C * 8));
    owned_.push_back(p);
    PB_HIP(hipMemcpy(p, ig->second->data, (size_t)C * 4, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy((float *)p + C, ib->second->data, (size_t)C * 4, hipMemcpyHostToDevice));
    out.g = (float *)p; out.b = (float *)p + C; out.C = C;
    return 0;
}

// mmcv ConvModule with a norm: conv without bias -> GroupNorm(32) -> ReLU; submodules `conv` and `gn`
int MaskEngine::load_conv_gn(const std::string &name, ConvGN &out, int ci_pad) {
    int r = pack_conv(name + ".conv", false, nullptr, nullptr, out.w, sa_, ci_pad);
    if (r) return r;
    return load_gn(name + ".gn", (int)tmap_[name + ".conv.weight"]->shape[0], out.gn);
}

int MaskEngine::load(const pb_tensor *w, int n) {
    PB_CHECK(cfg_.num_classes > 0 && cfg_.feat_channels % 32 == 0 && cfg_.mask_feat_channels % 32 == 0 &&
                 cfg_.mask_out_channels == 256 && cfg_.stacked_convs >= 1 && cfg_.nms_pre > 0 && cfg_.nms_pre <= 512 &&
                 cfg_.max_per_img > 0 && cfg_.max_batch >= 1,
             PB_ERR_ARG, "mask_mmdet: unsupported configuration");
    PB_CHECK(cfg_.precision == PB_PREC_F16 || cfg_.precision == PB_PREC_SPLIT, PB_ERR_ARG, "mask_mmdet: precision %d unknown", cfg_.precision);
    split_w_ = cfg_.precision == PB_PREC_SPLIT;
    // round 3: PB_PREC_SPLIT splits the activations too - every feature map is [hi (C) | lo (C)] fp16 pairs per pixel, the weights are packed
    // per tap as [w_hi | w_hi | w_lo] (engine_base.h pack sa) and every GEMM / convolution runs a_hi w_hi + a_lo w_hi + a_hi w_lo into one fp32
    // accumulator.  PB_MASK_SA=0 keeps single-fp16 activations (the round-2 schedule) for A/B.
    sa_ = split_w_ && pb_env_int("PB_MASK_SA", 1) ? 1 : 0;
    int r0 = begin_load(w, n);
    if (r0) return r0;
    for (int l = 0; l < 5; ++l) {
        PB_HIP(hipStreamCreateWithFlags(&ls_[l], hipStreamNonBlocking));
        PB_HIP(hipEventCreateWithFlags(&ev_join_[l], hipEventDisableTiming));
    }
    PB_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
    int r;
    std::vector<float> sc, sf;
    {   // stem 7x7 / stride 2 (resnet.py:560-580, no bias) + bn1: on the 4 x 4 space-to-depth image (mask_prep) it is a 3x3 / stride 1
        // convolution whose 4 x 64 output channels are the 2 x 2 output pixels of a block (pixel-shuffle epilogue), cf. raft_engine.hip:
        // W2[(sy, sx, o)][(ty, tx)][(dy, dx, c)] = w[o][c][ky][kx],  ky = 4 (ty - 1) + dy - 2 sy + 3 (same in x), zero outside 0..6
        auto iw = tmap_.find("backbone.conv1.weight");
        PB_CHECK(iw != tmap_.end(), PB_ERR_ARG, "missing backbone.conv1.weight");
        if ((r = fold_bn("backbone.bn1", 64, sc, sf))) return r;
        const float *wt = (const float *)iw->second->data;
        std::vector<float> g((size_t)256 * 576, 0.f), bb(256);
        for (int sy = 0; sy < 2; ++sy)
            for (int sx = 0; sx < 2; ++sx)
                for (int o = 0; o < 64; ++o) {
                    const int nn = (sy * 2 + sx) * 64 + o;
                    bb[nn] = sf[o];
                    for (int ty = 0; ty < 3; ++ty)
                        for (int tx = 0; tx < 3; ++tx)
                            for (int dy = 0; dy < 4; ++dy)
                                for (int dx = 0; dx < 4; ++dx) {
                                    const int ky = 4 * (ty - 1) + dy - 2 * sy + 3, kx = 4 * (tx - 1) + dx - 2 * sx + 3;
                                    if (ky < 0 || ky > 6 || kx < 0 || kx > 6) continue;
                                    for (int c = 0; c < 3; ++c)
                                        g[(size_t)nn * 576 + (ty * 3 + tx) * 64 + (dy * 4 + dx) * 4 + c] = wt[((size_t)o * 3 + c) * 49 + ky * 7 + kx] * sc[o];
                                }
                }
        if ((r = pack(g.data(), 256, 576, 576, stem_, bb.data(), 9, sa_))) return r;
        stem_.Kreal = 147;
    }
    int inpl = 64;
    for (int li = 0; li < 4; ++li) {
        const int planes = 64 << li;
        blocks_[li].resize(cfg_.blocks[li]);
        for (int b = 0; b < cfg_.blocks[li]; ++b) {
            Bneck &B = blocks_[li][b];
            const std::string p = "backbone.layer" + std::to_string(li + 1) + "." + std::to_string(b);
            B.planes = planes; B.inpl = inpl;
            B.stride = (b == 0 && li > 0) ? 2 : 1;              // style 'pytorch': the 3x3 carries the stride
            if ((r = fold_bn(p + ".bn1", planes, sc, sf)) || (r = pack_conv(p + ".conv1", false, sc.data(), sf.data(), B.c1, sa_))) return r;
            if ((r = fold_bn(p + ".bn2", planes, sc, sf)) || (r = pack_conv(p + ".conv2", false, sc.data(), sf.data(), B.c2, sa_))) return r;
            if ((r = fold_bn(p + ".bn3", planes * 4, sc, sf)) || (r = pack_conv(p + ".conv3", false, sc.data(), sf.data(), B.c3, sa_))) return r;
            B.has_ds = b == 0;
            if (B.has_ds) {
                if ((r = fold_bn(p + ".downsample.1", planes * 4, sc, sf)) ||
                    (r = pack_conv(p + ".downsample.0", false, sc.data(), sf.data(), B.ds, sa_)))
                    return r;
            }
            inpl = planes * 4;
        }
    }
    for (int i = 0; i < 4; ++i) {
        if ((r = pack_conv("neck.lateral_convs." + std::to_string(i) + ".conv", true, nullptr, nullptr, lat_[i], sa_))) return r;
        if ((r = pack_conv("neck.fpn_convs." + std::to_string(i) + ".conv", true, nullptr, nullptr, fpnc_[i], sa_))) return r;
    }
    const std::string h = "mask_head.mask_feature_head.";
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < std::max(i, 1); ++j)
            if ((r = load_conv_gn(h + "convs_all_levels." + std::to_string(i) + ".conv" + std::to_string(j), mfc_[i][j]))) return r;
    if ((r = load_conv_gn(h + "conv_pred", mfpred_))) return r;
    kconv_.resize(cfg_.stacked_convs);
    cconv_.resize(cfg_.stacked_convs);
    for (int i = 0; i < cfg_.stacked_convs; ++i) {
        if ((r = load_conv_gn("mask_head.kernel_convs." + std::to_string(i), kconv_[i]))) return r;
        // cate_feat = kernel_feat[:, :-2]: a channel slice of a split map has its residuals at +320, not +256, which the K wrap of the GEMM
        // cannot express - the first cls conv is packed for all 320 channels with zero weights on the coordinates instead
        if ((r = load_conv_gn("mask_head.cls_convs." + std::to_string(i), cconv_[i], i == 0 && sa_ ? 320 : 0))) return r;
    }
    if ((r = pack_conv("mask_head.conv_cls", true, nullptr, nullptr, conv_cls_, sa_))) return r;
    conv_cls_.N = (int)round_up(conv_cls_.N, 8);             // the GEMM writes 8-column groups; the pad rows are zero
    if ((r = pack_conv("mask_head.conv_kernel", true, nullptr, nullptr, conv_kernel_, sa_))) return r;
    PB_CHECK(conv_kernel_.N == cfg_.mask_out_channels && mfpred_.w.N == cfg_.mask_out_channels, PB_ERR_ARG,
             "mask_mmdet: kernel / mask feature widths do not match the configuration");
    tmap_.clear();
    PB_HIP(hipDeviceSynchronize());
    return 0;
}

int MaskEngine::prepare(int n, int H, int W) {
    const int B = std::min(n, cfg_.max_batch);
    if (B <= pB_ && H == pH_ && W == pW_) return 0;
    PB_HIP(hipStreamSynchronize(stream));
    pB_ = 0; pH_ = 0; pW_ = 0;          // (a failure below must not leave the old plan's key on a half-written plan: raft_engine.hip prepare)
    {   // the post-processing scratch is sized by the mask-feature resolution: drop it when the geometry changes
        void **post[] = {(void **)&pk_, (void **)&bits_, (void **)&plog_, (void **)&pstat_, (void **)&inter_, (void **)&sig_,
                         (void **)&nmsf_, (void **)&pidx_, (void **)&nmsi_, (void **)&use_};
        for (auto b : post)
            if (*b) { PB_HIP(hipFree(*b)); *b = nullptr; }
        post_cap_ = 0;
        post_frames_ = 0;
    }
    net_size(cfg_, H, W, &nh_, &nw_, &Hp_, &Wp_);
    PB_CHECK(Hp_ >= 64 && Wp_ >= 64, PB_ERR_ARG, "mask_mmdet: network input %dx%d is too small", Hp_, Wp_);
    lh_[5] = Hp_ / 2; lw_[5] = Wp_ / 2;
    for (int i = 0; i < 4; ++i) { lh_[i] = Hp_ / (4 << i); lw_[i] = Wp_ / (4 << i); }
    lh_[4] = (lh_[3] - 1) / 2 + 1; lw_[4] = (lw_[3] - 1) / 2 + 1;
    pts_ = 0;
    int gmax = 0;
    for (int l = 0; l < 5; ++l) { goff_[l] = pts_; pts_ += cfg_.num_grids[l] * cfg_.num_grids[l]; gmax = std::max(gmax, cfg_.num_grids[l]); }
    goff_[5] = pts_;
    const int fc = cfg_.feat_channels, mfc = cfg_.mask_feat_channels, Cp = conv_cls_.N;
    const size_t slack = 1 << 20;
    const int sp = 1 + sa_;                                 // fp16 parts per element of a feature map
    auto rows = [&](int i) { return (size_t)round_up((int64_t)B * lh_[i] * lw_[i], 256) * sp; };
    for (int pass = 0; pass < 2; ++pass) {
        planning_ = pass == 0;
        arena_off_ = 0;
        xt_ = (int *)carve((size_t)nw_ * 16); yt_ = (int *)carve((size_t)nh_ * 16);
        img_ = (f16 *)carve((size_t)B * Hp_ * Wp_ * 8 * sp + slack);
        chw_ = debug ? (float *)carve((size_t)B * 3 * Hp_ * Wp_ * 4) : nullptr;
        stem_out_ = (f16 *)carve(rows(5) * 64 * 2);
        pool_ = (f16 *)carve(rows(0) * 64 * 2 + slack);
        for (int s = 0; s < 4; ++s) {
            const int p = 64 << s, in = s == 0 ? 0 : s - 1;
            sx_[s][0] = (f16 *)carve(rows(s) * p * 4 * 2 + slack);
            sx_[s][1] = (f16 *)carve(rows(s) * p * 4 * 2 + slack);
            st1_[s] = (f16 *)carve(rows(in) * p * 2 + slack);
            st2_[s] = (f16 *)carve(rows(s) * p * 2 + slack);
            sds_[s] = (f16 *)carve(rows(s) * p * 4 * 2 + slack);
        }
        for (int i = 0; i < 4; ++i) { latb_[i] = (f16 *)carve(rows(i) * 256 * 2 + slack); p_[i] = (f16 *)carve(rows(i) * 256 * 2 + slack); }
        p_[4] = (f16 *)carve(rows(4) * 256 * 2 + slack);
        mt_[0] = (f16 *)carve(rows(0) * 256 * 2 + slack);
        mg_[0] = (f16 *)carve(rows(0) * mfc * 2 + slack); mg_[1] = (f16 *)carve(rows(0) * mfc * 2 + slack);
        macc_ = (f16 *)carve(rows(0) * mfc * 2 + slack);
        mf_ = (f16 *)carve(rows(0) / sp * (sa_ ? 3 : 1) * 256 * 2 + slack);       // split: [hi | hi | lo] rows (B operand of the dynamic convolution)
        p5cc_ = (f16 *)carve(rows(3) * 320 * 2 + slack);
        for (int l = 0; l < 5; ++l) {                        // each level's two branches run on their own stream
            const int fl = l == 0 ? 1 : (l == 4 ? 3 : l);     // resolution after resize_feats
            const size_t grows = (size_t)round_up((int64_t)B * cfg_.num_grids[l] * cfg_.num_grids[l], 256) * sp;
            fcc_[l] = (f16 *)carve(rows(fl) * 320 * 2 + slack);
            rs_[l] = (l == 0 || l == 4) ? (f16 *)carve(rows(fl) * 256 * 2 + slack) : nullptr;
            grid_[l] = (f16 *)carve(grows * 320 * 2 + slack);
            for (auto &b : hk_[l]) b = (f16 *)carve(grows * fc * 2 + slack);
            gstl_[l] = (float *)carve((size_t)B * gn_chunks(cfg_.num_grids[l] * cfg_.num_grids[l]) * 512 * 2 * 4); gaffl_[l] = (float *)carve((size_t)B * 512 * 2 * 4);
        }
        kp_ = (float *)carve((size_t)B * pts_ * 256 * 4 + slack);
        cl_ = (float *)carve((size_t)B * pts_ * Cp * 4 + slack);
        cs_ = (float *)carve((size_t)B * pts_ * Cp * 4 + slack);
        gst_ = (float *)carve((size_t)B * gn_chunks(lh_[0] * lw_[0]) * 512 * 2 * 4); gaff_ = (float *)carve((size_t)B * 512 * 2 * 4);
        if (pass == 0) {
            const int rc = commit_arena("mask");
            if (rc) return rc;
        }
    }
    std::vector<int> xt, yt;
    linear_taps_u8(W, nw_, false, xt);
    linear_taps_u8(H, nh_, true, yt);
    PB_HIP(hipMemcpyAsync(xt_, xt.data(), xt.size() * 4, hipMemcpyHostToDevice, stream));
    PB_HIP(hipMemcpyAsync(yt_, yt.data(), yt.size() * 4, hipMemcpyHostToDevice, stream));
    PB_HIP(hipStreamSynchronize(stream));
    pB_ = B; pH_ = H; pW_ = W;
    return 0;
}

int MaskEngine::conv_f32(const f16 *in, int cC, int cLd, int n, int H, int W, const PackedW &w, float *out, int ldo) {
    GemmArgs a;
    a.A = in; a.N = w.N;
    a.cH = H; a.cW = W; a.cC = cC; a.cLd = L(cLd); a.cKW = 3; a.cStride = 1; a.cPad = 1; a.cPadX = 1; a.cOH = H; a.cOW = W;
    a.M = n * H * W;
    a.out32 = out; a.ldo = ldo; a.scale = 1.f;
    set_weights(a, w, true);
    PB_CHECK(w.K == 9 * a.cC, PB_ERR_STATE, "conv_f32: packed K %d != 9*%d", w.K, a.cC);
    tic(F_CONV, 2.0 * a.M * (double)a.N * w.Kreal, 0);
    int r = launch_gemm(cur_, A_CONV, EPI_F32, TILE_128, a);
    if (timer.enabled && !r && !open_.empty()) timer.recs[open_.back()].name = pb_gemm_last_kernel();
    toc();
    return r;
}

int MaskEngine::conv_gn_relu(const f16 *in, int cC, int cLd, int n, int H, int W, int k, const ConvGN &c, f16 *tmp, f16 *out,
                             int ldo, int dup) {
    const int N = c.w.N;
    int r = k == 1 ? dense(in, L(cLd), (int64_t)n * H * W, c.w, tmp, L(N), ACT_NONE, nullptr, 0, -1, lo(N))
                   : conv(in, cC, L(cLd), n, H, W, k, k, 1, c.w, tmp, L(N), ACT_NONE, 0, nullptr, nullptr, lo(N));
    if (r) return r;
    tic(F_ELT, 0, (double)n * H * W * N * 6 * (1 + sa_));
    // dup (split only): rows [hi | hi | lo] instead of [hi | lo]
    r = launch_gn_relu(cur_, tmp, out, n, H * W, c.gn.C, L(N), dup && sa_ ? 3 * ldo : L(ldo), 32, c.gn.g, c.gn.b, gst_cur_, gaff_cur_, lo(N),
                       dup && sa_ ? 2 * ldo : lo(ldo), dup && sa_ ? ldo : 0);
    toc();
    return r;
}

int MaskEngine::backbone(int n) {
    int r;
    const int H2 = lh_[5], W2 = lw_[5];
    {
        GemmArgs a;
        a.A = img_; a.N = 256;
        a.cH = Hp_ / 4; a.cW = Wp_ / 4; a.cC = 64; a.cLd = L(64); a.cKW = 3; a.cStride = 1; a.cPad = 1; a.cPadX = 1;
        set_weights(a, stem_, true);
        a.cOH = a.cH; a.cOW = a.cW; a.M = n * a.cH * a.cW;
        a.out = stem_out_; a.ldo = L(64); a.lo_off = lo(64); a.act = ACT_RELU;
        a.ps_h = a.cH; a.ps_w = a.cW; a.ps_s = 2; a.ps_co = 64;
        tic(F_CONV, 2.0 * n * H2 * W2 * 64.0 * 147, 0);
        r = launch_gemm(cur_, A_CONV, EPI_PIXSHUF, TILE_AUTO, a);
        if (timer.enabled && !r && !open_.empty()) timer.recs[open_.back()].name = pb_gemm_last_kernel();
        toc();
        if (r) return r;
    }
    tic(F_ELT, 0, 0);
    r = launch_maxpool3x3s2(stream, stem_out_, pool_, n, H2, W2, 64, sa_);
    toc();
    if (r) return r;
    const f16 *x = pool_;
    int hi = lh_[0], wi = lw_[0];
    for (int s = 0; s < 4; ++s) {
        for (size_t b = 0; b < blocks_[s].size(); ++b) {
            const Bneck &B = blocks_[s][b];
            const int ho = (hi - 1) / B.stride + 1, wo = (wi - 1) / B.stride + 1, p = B.planes;
            if ((r = dense(x, L(B.inpl), (int64_t)n * hi * wi, B.c1, st1_[s], L(p), ACT_RELU, nullptr, 0, -1, lo(p)))) return r;
            if ((r = conv(st1_[s], p, L(p), n, hi, wi, 3, 3, B.stride, B.c2, st2_[s], L(p), ACT_RELU, 0, nullptr, nullptr, lo(p)))) return r;
            const f16 *idt = x;
            if (B.has_ds) {
                if (B.stride == 1) r = dense(x, L(B.inpl), (int64_t)n * hi * wi, B.ds, sds_[s], L(4 * p), ACT_NONE, nullptr, 0, -1, lo(4 * p));
                else r = conv(x, B.inpl, L(B.inpl), n, hi, wi, 1, 1, B.stride, B.ds, sds_[s], L(4 * p), ACT_NONE, 0, nullptr, nullptr, lo(4 * p));
                if (r) return r;
                idt = sds_[s];
            }
            f16 *out = sx_[s][b & 1];
            if ((r = dense(st2_[s], L(p), (int64_t)n * ho * wo, B.c3, out, L(4 * p), ACT_RELU, idt, 0, -1, lo(4 * p)))) return r;
            x = out; hi = ho; wi = wo;
        }
        PB_CHECK(hi == lh_[s] && wi == lw_[s], PB_ERR_STATE, "backbone: stage %d size %dx%d != %dx%d", s, hi, wi, lh_[s], lw_[s]);
        c_[s] = x;
        stages_["c" + std::to_string(s + 2)] = Stage{x, 1, 0, 256 << s, lh_[s], lw_[s], L(256 << s), lo(256 << s)};
    }
    return 0;
}

int MaskEngine::neck(int n) {
    int r;
    for (int i = 0; i < 4; ++i)
        if ((r = dense(c_[i], L(256 << i), (int64_t)n * lh_[i] * lw_[i], lat_[i], latb_[i], L(256), ACT_NONE, nullptr, 0, -1, lo(256)))) return r;
    for (int i = 3; i > 0; --i) {
        tic(F_ELT, 0, 0);
        r = launch_nearest_add(stream, latb_[i - 1], latb_[i], n, lh_[i - 1], lw_[i - 1], lh_[i], lw_[i], 256, sa_);
        toc();
        if (r) return r;
    }
    for (int i = 0; i < 4; ++i)
        if ((r = conv(latb_[i], 256, L(256), n, lh_[i], lw_[i], 3, 3, 1, fpnc_[i], p_[i], L(256), ACT_NONE, 0, nullptr, nullptr, lo(256)))) return r;
    tic(F_ELT, 0, 0);
    r = launch_subsample2(stream, p_[3], p_[4], n, lh_[3], lw_[3], L(256));         // whole rows: both parts of a split map
    toc();
    for (int i = 0; i < 5; ++i) stages_["p" + std::to_string(i + 2)] = Stage{p_[i], 1, 0, 256, lh_[i], lw_[i], L(256), lo(256)};
    return r;
}

int MaskEngine::head(int n) {
    int r;
    const int mfc = cfg_.mask_feat_channels;
    // ---- MaskFeatModule.forward (solov2_head.py:134-150) ----
    if ((r = conv_gn_relu(p_[0], 256, 256, n, lh_[0], lw_[0], 3, mfc_[0][0], mt_[0], macc_, mfc))) return r;
    for (int i = 1; i < 4; ++i) {
        const f16 *x = p_[i];
        int cC = 256, h = lh_[i], w = lw_[i];
        if (i == 3) {
            tic(F_ELT, 0, 0);
            r = launch_coord_concat(stream, p_[3], p5cc_, n, h, w, 256, L(256), lo(256));
            toc();
            if (r) return r;
            x = p5cc_; cC = 320;
        }
        for (int j = 0; j < i; ++j) {
            if ((r = conv_gn_relu(x, cC, cC, n, h, w, 3, mfc_[i][j], mt_[0], mg_[0], mfc))) return r;
            const bool last = j == i - 1;
            tic(F_ELT, 0, 0);
            r = launch_bilinear(stream, mg_[0], last ? macc_ : mg_[1], n, h, w, 2 * h, 2 * w, mfc, L(mfc), L(mfc), last ? 1 : 0, lo(mfc), lo(mfc));
            toc();
            if (r) return r;
            x = mg_[1]; cC = mfc; h *= 2; w *= 2;
        }
        PB_CHECK(h == lh_[0] && w == lw_[0], PB_ERR_STATE, "mask feature level %d ends at %dx%d", i, h, w);
    }
    if ((r = conv_gn_relu(macc_, mfc, mfc, n, lh_[0], lw_[0], 1, mfpred_, mt_[0], mf_, 256, 1))) return r;
    stages_["mask_feats"] = Stage{mf_, 1, 0, 256, lh_[0], lw_[0], sa_ ? 768 : 256, sa_ ? 512 : 0};

    return 0;
}

// resize_feats + the two prediction branches of one level (solov2_head.py:253-292), on that level's stream
int MaskEngine::head_level(int n, int lvl) {
    int r;
    const int fc = cfg_.feat_channels, Cp = conv_cls_.N;
    const int g = cfg_.num_grids[lvl];
    f16 *fcc = fcc_[lvl], *rs = rs_[lvl], *grid = grid_[lvl], **hk = hk_[lvl];
    const f16 *src = p_[lvl];
    int fh = lh_[lvl], fw = lw_[lvl];
    if (lvl == 0 || lvl == 4) {
        const int th = lvl == 0 ? lh_[1] : lh_[3], tw = lvl == 0 ? lw_[1] : lw_[3];
        tic(F_ELT, 0, 0);
        r = launch_bilinear(cur_, src, rs, n, fh, fw, th, tw, 256, L(256), L(256), 0, lo(256), lo(256));
        toc();
        if (r) return r;
        src = rs; fh = th; fw = tw;
    }
    tic(F_ELT, 0, 0);
    r = launch_coord_concat(cur_, src, fcc, n, fh, fw, 256, L(256), lo(256));
    if (!r) r = launch_bilinear(cur_, fcc, grid, n, fh, fw, g, g, 320, L(320), L(320), 0, lo(320), lo(320));
    toc();
    if (r) return r;
    const f16 *x = grid;
    int cC = 320, cLd = 320;
    for (int i = 0; i < cfg_.stacked_convs; ++i) {
        f16 *o = hk[1 + (i & 1)];
        if ((r = conv_gn_relu(x, cC, cLd, n, g, g, 3, kconv_[i], hk[0], o, fc))) return r;
        x = o; cC = cLd = fc;
    }
    float *kp = kp_ + (int64_t)n * goff_[lvl] * 256;
    if ((r = conv_f32(x, fc, fc, n, g, g, conv_kernel_, kp, 256))) return r;
    x = grid; cC = sa_ ? 320 : 256; cLd = 320;                       // cate_feat = kernel_feat[:, :-2] (split maps: zero weights on the coordinates, load())
    for (int i = 0; i < cfg_.stacked_convs; ++i) {
        f16 *o = hk[1 + (i & 1)];
        if ((r = conv_gn_relu(x, cC, cLd, n, g, g, 3, cconv_[i], hk[0], o, fc))) return r;
        x = o; cC = cLd = fc;
    }
    float *cl = cl_ + (int64_t)n * goff_[lvl] * Cp;
    if ((r = conv_f32(x, fc, fc, n, g, g, conv_cls_, cl, Cp))) return r;
    tic(F_PP, 0, (double)n * g * g * Cp * 8);
    r = launch_cls_points_nms(cur_, cl, cs_, n, pts_, goff_[lvl], g, Cp);
    toc();
    if (r) return r;
    stages_["kernel_pred" + std::to_string(lvl)] = Stage{kp, 3, 0, 256, g, g, 256, 0};
    stages_["cls_logit" + std::to_string(lvl)] = Stage{cl, 3, 0, cfg_.num_classes, g, g, Cp, 0};
    return 0;
}

int MaskEngine::ensure_post(size_t cands, int frames) {
    const size_t HW4 = (size_t)lh_[0] * lw_[0];
    const size_t need = round_up((int64_t)std::max<size_t>(cands, 256), 256);
    PB_CHECK(need <= (1u << 17), PB_ERR_MEMORY, "mask_mmdet: %zu candidate cells over score_thr in one chunk", cands);
    if (need > post_cap_) {
        PB_HIP(hipStreamSynchronize(stream));
        void **bufs[] = {(void **)&pk_, (void **)&plog_, (void **)&pstat_, (void **)&pidx_};
        for (auto b : bufs)
            if (*b) { PB_HIP(hipFree(*b)); *b = nullptr; }
        PB_HIP(hipMalloc((void **)&pk_, (need + 256) * 256 * 2 * (1 + sa_)));
        PB_HIP(hipMalloc((void **)&plog_, need * HW4 * 4));
        PB_HIP(hipMalloc((void **)&pstat_, need * 2 * 4));
        PB_HIP(hipMalloc((void **)&pidx_, need * 4));
        post_cap_ = need;
    }
    if (frames > post_frames_) {        // per-frame Matrix-NMS / final-mask scratch
        PB_HIP(hipStreamSynchronize(stream));
        void **bufs[] = {(void **)&bits_, (void **)&inter_, (void **)&nmsf_, (void **)&nmsi_, (void **)&sig_, (void **)&use_};
        for (auto b : bufs)
            if (*b) { PB_HIP(hipFree(*b)); *b = nullptr; }
        PB_HIP(hipMalloc((void **)&bits_, (size_t)frames * 512 * (HW4 / 64) * 8));
        PB_HIP(hipMalloc((void **)&inter_, (size_t)frames * 512 * 512 * 4));
        PB_HIP(hipMalloc((void **)&nmsf_, (size_t)frames * 5 * 512 * 4));
        PB_HIP(hipMalloc((void **)&nmsi_, (size_t)frames * 3 * 512 * 4));
        PB_HIP(hipMalloc((void **)&sig_, (size_t)frames * cfg_.max_per_img * HW4 * 4));
        PB_HIP(hipMalloc((void **)&use_, (size_t)frames * 512));
        post_frames_ = frames;
    }
    return 0;
}

// _get_results_single (solov2_head.py:647-766) + format_results + the band's accumulation, for all frames of a chunk in three
// passes so that the host decides once per pass (three stream syncs per chunk, not per frame):
//   A  cells over score_thr -> gather their kernels -> dynamic convolution -> mask area / maskness sums
//   B  area filter, maskness, sort, nms_pre -> bit masks -> pairwise intersections -> Matrix-NMS decay
//   C  filter_thr, sort, max_per_img -> sigmoid rows -> upsample x4, crop, resize, threshold, accumulate
int MaskEngine::post_chunk(int n, int first, float confidence, const std::vector<uint8_t> &keep_class, uint8_t *mask_out) {
    const int C = cfg_.num_classes, Cp = conv_cls_.N;
    const int fh = lh_[0], fw = lw_[0], HW4 = fh * fw;
    const int64_t opix = (int64_t)pH_ * pW_;
    struct Cand { int cell, label; float score; };
    struct Kept { int row, label; float score, area; };
    struct Fin { int row, label; float score; };
    struct Frame {
        std::vector<Cand> cand;
        std::vector<float> stride_of, st, ns;
        std::vector<Kept> kept;
        size_t off = 0;             // first row of this frame in pk_ / plog_ / pstat_
        bool done = false;
    };
    std::vector<Frame> fr(n);
    int r;
    auto finish_empty = [&](int b) -> int {
        fr[b].done = true;
        PB_HIP(hipMemsetAsync(mask_out + (int64_t)(first + b) * opix * 3, 0, (size_t)opix * 3, stream));
        return 0;
    };
    // ---------------- pass A ----------------
    size_t total = 0;
    for (int b = 0; b < n; ++b) {
        Frame &f = fr[b];
        results_[first + b] = Instances();
        const float *sc = h_scores_.data() + (int64_t)b * pts_ * Cp;
        for (int p = 0; p < pts_; ++p)          // nonzero() order: cell-major, class-minor
            for (int c = 0; c < C; ++c)
                if (sc[(int64_t)p * Cp + c] > cfg_.score_thr) f.cand.push_back({p, c, sc[(int64_t)p * Cp + c]});
        results_[first + b].candidates = (int)f.cand.size();
        f.off = total;
        total += round_up((int64_t)f.cand.size(), 8);
    }
    if ((r = ensure_post(total, n))) return r;
    std::vector<int> rows(total, 0);
    for (int b = 0; b < n; ++b) {
        Frame &f = fr[b];
        f.stride_of.resize(f.cand.size());
        for (size_t i = 0; i < f.cand.size(); ++i) {
            int lvl = 0;
            while (f.cand[i].cell >= goff_[lvl + 1]) ++lvl;
            rows[f.off + i] = n * goff_[lvl] + b * (goff_[lvl + 1] - goff_[lvl]) + (f.cand[i].cell - goff_[lvl]);   // level-major kernel rows
            f.stride_of[i] = (float)cfg_.strides[lvl];
        }
    }
    if (total) {
        PB_HIP(hipMemcpyAsync(pidx_, rows.data(), total * 4, hipMemcpyHostToDevice, stream));
        tic(F_PP, 0, (double)total * 256 * 6);
        r = launch_gather_rows_f16(stream, kp_, pidx_, pk_, (int)total, (int)total, 256, sa_);
        toc();
        if (r) return r;
    }
    for (int b = 0; b < n; ++b) {
        Frame &f = fr[b];
        const int K = (int)f.cand.size();
        if (!K) { if ((r = finish_empty(b))) return r; continue; }
        GemmArgs a;         // dynamic convolution: logits[k][pixel] = <kernel_k, mask_feats[pixel]>
        // split: kernels [k_hi | k_lo] x features [f_hi | f_hi | f_lo] with the K wrap of gemm.h - k_hi f_hi + k_lo f_hi + k_hi f_lo
        a.A = pk_ + f.off * L(256); a.lda = L(256); a.M = K; a.W = mf_ + (int64_t)b * HW4 * (sa_ ? 768 : 256); a.K = sa_ ? 768 : 256; a.N = HW4;
        if (sa_) a.kwrap = 8;
        a.out32 = plog_ + f.off * HW4; a.ldo = HW4; a.scale = 1.f; a.zero = zero_;
        tic(F_GEMM, 2.0 * K * (double)HW4 * 256, 0, sa_ ? 3.0 : 1.0);
        r = launch_gemm(stream, A_DENSE, EPI_F32, TILE_AUTO, a);
        if (timer.enabled && !r && !open_.empty()) timer.recs[open_.back()].name = pb_gemm_last_kernel();
        toc();
        if (r) return r;
        tic(F_PP, 0, (double)K * HW4 * 4);
        r = launch_mask_stats(stream, plog_ + f.off * HW4, K, HW4, HW4, cfg_.mask_thr, pstat_ + f.off * 2);
        toc();
        if (r) return r;
        f.st.resize((size_t)K * 2);
        PB_HIP(hipMemcpyAsync(f.st.data(), pstat_ + f.off * 2, (size_t)K * 8, hipMemcpyDeviceToHost, stream));
    }
    PB_HIP(hipStreamSynchronize(stream));
    // ---------------- pass B ----------------
    std::vector<std::vector<int>> li(n);
    std::vector<std::vector<float>> lf(n);
    for (int b = 0; b < n; ++b) {
        Frame &f = fr[b];
        if (f.done) continue;
        const int K = (int)f.cand.size();
        for (int i = 0; i < K; ++i)             // keep = sum_masks > strides; cls_scores *= maskness
            if (f.st[(size_t)i * 2] > f.stride_of[i])
                f.kept.push_back({i, f.cand[i].label, f.cand[i].score * (f.st[(size_t)i * 2 + 1] / f.st[(size_t)i * 2]), f.st[(size_t)i * 2]});
        if (f.kept.empty()) { if ((r = finish_empty(b))) return r; continue; }
        std::stable_sort(f.kept.begin(), f.kept.end(), [](const Kept &x, const Kept &y) { return x.score > y.score; });
        if ((int)f.kept.size() > cfg_.nms_pre) f.kept.resize(cfg_.nms_pre);
        const int n2 = (int)f.kept.size();
        li[b].assign(1024, 0);
        lf[b].assign(1024, 0.f);
        for (int i = 0; i < n2; ++i) {
            li[b][i] = (int)f.off + f.kept[i].row; li[b][512 + i] = f.kept[i].label;
            lf[b][i] = f.kept[i].area; lf[b][512 + i] = f.kept[i].score;
        }
        int *ni = nmsi_ + (size_t)b * 3 * 512;
        float *nf = nmsf_ + (size_t)b * 5 * 512;
        unsigned long long *bits = bits_ + (size_t)b * 512 * (HW4 / 64);
        float *inter = inter_ + (size_t)b * 512 * 512;
        PB_HIP(hipMemcpyAsync(ni, li[b].data(), 1024 * 4, hipMemcpyHostToDevice, stream));
        PB_HIP(hipMemcpyAsync(nf, lf[b].data(), 1024 * 4, hipMemcpyHostToDevice, stream));
        // binary masks as bit rows; pairwise intersections by popcount (matrix_nms.py:66-67), then the decay
        tic(F_PP, 0, (double)n2 * HW4 * 4 + (double)n2 * n2 * (HW4 / 8));
        r = launch_bitpack_rows(stream, plog_, HW4, ni, n2, HW4, cfg_.mask_thr, bits);
        if (!r) r = launch_mask_intersections(stream, bits, n2, HW4 / 64, inter, 512);
        if (!r) r = launch_matrix_nms(stream, inter, 512, nf, ni + 512, nf + 512, n2, cfg_.sigma, nf + 1536, nf + 1024);
        toc();
        if (r) return r;
        f.ns.resize(n2);
        PB_HIP(hipMemcpyAsync(f.ns.data(), nf + 1024, (size_t)n2 * 4, hipMemcpyDeviceToHost, stream));
    }
    PB_HIP(hipStreamSynchronize(stream));
    // ---------------- pass C ----------------
    std::vector<std::vector<int>> frow(n);
    std::vector<std::vector<uint8_t>> use(n);
    for (int b = 0; b < n; ++b) {
        Frame &f = fr[b];
        if (f.done) continue;
        Instances &res = results_[first + b];
        std::vector<Fin> fin;
        for (size_t i = 0; i < f.kept.size(); ++i)
            if (!(cfg_.filter_thr > 0.f) || f.ns[i] >= cfg_.filter_thr) fin.push_back({(int)f.off + f.kept[i].row, f.kept[i].label, f.ns[i]});
        if (fin.empty()) { if ((r = finish_empty(b))) return r; continue; }
        std::stable_sort(fin.begin(), fin.end(), [](const Fin &x, const Fin &y) { return x.score > y.score; });
        if ((int)fin.size() > cfg_.max_per_img) fin.resize(cfg_.max_per_img);
        const int n3 = (int)fin.size();
        // the band (mask_mmdet.py:43-49,139-147): classes in the keep list, score > 0.5 (getTotalMasks' default) and > --confidence
        frow[b].resize(n3);
        use[b].resize(n3);
        for (int i = 0; i < n3; ++i) {
            frow[b][i] = fin[i].row;
            use[b][i] = keep_class[fin[i].label] && fin[i].score > 0.5f && fin[i].score > confidence;
            res.scores.push_back(fin[i].score);
            res.labels.push_back(fin[i].label);
        }
        int *ni = nmsi_ + (size_t)b * 3 * 512 + 1024;
        uint8_t *us = use_ + (size_t)b * 512;
        float *sig = sig_ + (size_t)b * cfg_.max_per_img * HW4;
        PB_HIP(hipMemcpyAsync(ni, frow[b].data(), (size_t)n3 * 4, hipMemcpyHostToDevice, stream));
        PB_HIP(hipMemcpyAsync(us, use[b].data(), (size_t)n3, hipMemcpyHostToDevice, stream));
        tic(F_PP, 0, (double)n3 * HW4 * 8);
        r = launch_sigmoid_rows(stream, plog_, HW4, ni, n3, HW4, sig);
        toc();
        if (r) return r;
        uint8_t *inst = nullptr;
        if (debug) {        // parity dumps only: one frame at a time through a shared buffer
            const size_t need = (size_t)n3 * opix;
            if (need > inst_bytes_) {
                PB_HIP(hipStreamSynchronize(stream));
                if (inst_) PB_HIP(hipFree(inst_));
                PB_HIP(hipMalloc((void **)&inst_, need));
                inst_bytes_ = need;
            }
            inst = inst_;
        }
        tic(F_PP, 0, (double)opix * (3 + 64.0 * n3));
        r = launch_band_accumulate(stream, sig, n3, fh, fw, nh_, nw_, pH_, pW_, cfg_.mask_thr, us, mask_out + (int64_t)(first + b) * opix * 3, inst);
        toc();
        if (r) return r;
        if (debug) {
            res.masks.resize((size_t)n3 * opix);
            PB_HIP(hipMemcpyAsync(res.masks.data(), inst_, (size_t)n3 * opix, hipMemcpyDeviceToHost, stream));
            PB_HIP(hipStreamSynchronize(stream));
        }
    }
    PB_HIP(hipStreamSynchronize(stream));       // the host index vectors of pass C go out of scope here
    return 0;
}

int MaskEngine::run_chunk(const uint8_t *frames, int n, int first, float confidence, const std::vector<uint8_t> &keep_class,
                          uint8_t *mask_out) {
    int r;
    last_n_ = n;
    tic(F_PP, 0, (double)n * ((double)pH_ * pW_ * 3 + (double)Hp_ * Wp_ * 8));
    r = launch_mask_prep(stream, frames, n, pH_, pW_, nh_, nw_, Hp_, Wp_, xt_, yt_, img_, chw_, sa_);
    toc();
    if (r) return r;
    if (chw_) stages_["input"] = Stage{chw_, 2, 0, 3, Hp_, Wp_, 0, 0};
    if ((r = backbone(n)) || (r = neck(n))) return r;
    // fork: the five levels' prediction branches only read the FPN outputs and are small (n * g^2 rows each), so they
    // run on their own streams beside the mask-feature branch on the main stream
    PB_HIP(hipEventRecord(ev_fork_, stream));
    for (int l = 0; l < 5 && !r; ++l) {
        PB_HIP(hipStreamWaitEvent(ls_[l], ev_fork_, 0));
        cur_ = ls_[l]; gst_cur_ = gstl_[l]; gaff_cur_ = gaffl_[l];
        r = head_level(n, l);
        cur_ = stream; gst_cur_ = gst_; gaff_cur_ = gaff_;
        if (!r) PB_HIP(hipEventRecord(ev_join_[l], ls_[l]));
    }
    if (r) return r;
    if ((r = head(n))) return r;
    for (int l = 0; l < 5; ++l) PB_HIP(hipStreamWaitEvent(stream, ev_join_[l], 0));
    const int Cp = conv_cls_.N;
    h_scores_.resize((size_t)n * pts_ * Cp);
    PB_HIP(hipMemcpyAsync(h_scores_.data(), cs_, h_scores_.size() * 4, hipMemcpyDeviceToHost, stream));
    PB_HIP(hipStreamSynchronize(stream));
    return post_chunk(n, first, confidence, keep_class, mask_out);
}

int MaskEngine::infer(const uint8_t *frames, int n, int H, int W, float confidence, const int32_t *keep, int n_keep,
                      uint8_t *mask_out) {
    PB_CHECK(frames && mask_out && n > 0 && H > 0 && W > 0, PB_ERR_ARG, "mask infer: bad arguments");
    PB_HIP(hipSetDevice(device));
    std::vector<uint8_t> keep_class(cfg_.num_classes, keep ? 0 : 1);
    for (int i = 0; i < n_keep && keep; ++i) {
        PB_CHECK(keep[i] >= 0 && keep[i] < cfg_.num_classes, PB_ERR_ARG, "mask infer: class id %d out of range", keep[i]);
        keep_class[keep[i]] = 1;
    }
    int r = prepare(n, H, W);
    if (r) return r;
    cur_ = stream; gst_cur_ = gst_; gaff_cur_ = gaff_;
    timer.reset();
    open_.clear();
    stages_.clear();
    results_.assign(n, Instances());
    auto cb = std::move(chunk_begin), ce = std::move(chunk_end);       // hooks are per call
    chunk_begin = nullptr; chunk_end = nullptr;
    for (int s = 0; s < n; s += pB_) {
        const int m = std::min(pB_, n - s);
        if (cb && (r = cb(s, m))) return r;
        if ((r = run_chunk(frames + (int64_t)s * H * W * 3, m, s, confidence, keep_class, mask_out))) return r;
        // the signed distance field is per frame (mask_kernels.hip sdf_*_kernel work on one frame's id image each), so with chunk hooks it is
        // applied chunk by chunk and a finished chunk can leave for the host while the next one runs
        if (ce && sdf_ncap_ > 0 && (r = sdf_green(mask_out + (int64_t)s * H * W * 3, m, H, W))) return r;
        if (ce && (r = ce(s, m))) return r;
    }
    if (!ce && sdf_ncap_ > 0) return sdf_green(mask_out, n, H, W);
    return 0;
}

int MaskEngine::set_sdf(const uint8_t *tab_out, const uint8_t *tab_in, int n_tab) {
    PB_HIP(hipSetDevice(device));
    if (n_tab <= 0) { sdf_ncap_ = 0; return 0; }
    PB_CHECK(tab_out && tab_in && n_tab >= 2, PB_ERR_ARG, "set_sdf: bad tables");
    PB_HIP(hipStreamSynchronize(stream));
    if (sdf_tab_) { PB_HIP(hipFree(sdf_tab_)); sdf_tab_ = nullptr; }
    PB_HIP(hipMalloc((void **)&sdf_tab_, (size_t)n_tab * 2));
    PB_HIP(hipMemcpy(sdf_tab_, tab_out, n_tab, hipMemcpyHostToDevice));
    PB_HIP(hipMemcpy(sdf_tab_ + n_tab, tab_in, n_tab, hipMemcpyHostToDevice));
    sdf_ncap_ = n_tab - 1;
    return 0;
}

int MaskEngine::sdf_green(uint8_t *masks, int n, int H, int W) {
    PB_CHECK(sdf_ncap_ > 0, PB_ERR_STATE, "sdf_green: no tables set (pb_mask_set_sdf)");
    PB_CHECK(masks && n > 0 && H > 0 && W > 0, PB_ERR_ARG, "sdf_green: bad arguments");
    PB_HIP(hipSetDevice(device));
    const size_t px = (size_t)n * H * W;
    if (px > sdf_px_) {
        PB_HIP(hipStreamSynchronize(stream));
        if (sdf_g_) { PB_HIP(hipFree(sdf_g_)); sdf_g_ = nullptr; }
        PB_HIP(hipMalloc((void **)&sdf_g_, round_up(2 * px, 16) + 4 * 4096));
        sdf_px_ = px;
    }
    PB_CHECK(n <= 4096, PB_ERR_ARG, "sdf_green: %d frames per call (at most 4096)", n);
    tic(F_PP, 0, (double)px * 14.0);
    const int r = launch_sdf_green(stream, masks, n, H, W, sdf_g_, sdf_g_ + px, (int *)(sdf_g_ + round_up(2 * sdf_px_, 16)), sdf_tab_,
                                   sdf_tab_ + sdf_ncap_ + 1, sdf_ncap_);
    toc();
    return r;
}

int64_t MaskEngine::get_stage(const char *name, float *out, int64_t cap, int64_t shape[4]) {
    auto it = stages_.find(name);
    PB_CHECK(it != stages_.end(), PB_ERR_ARG, "unknown stage '%s'", name);
    const Stage &s = it->second;
    PB_HIP(hipStreamSynchronize(stream));
    const int n = last_n_;
    const int64_t total = (int64_t)n * s.c * s.h * s.w;
    shape[0] = n; shape[1] = s.c; shape[2] = s.h; shape[3] = s.w;
    PB_CHECK(total <= cap, PB_ERR_ARG, "stage buffer too small");
    if (s.kind == 2) {          // fp32 NCHW already
        PB_HIP(hipMemcpy(out, s.ptr, total * 4, hipMemcpyDeviceToHost));
        return total;
    }
    if (s.kind == 3) {          // fp32 rows [n*h*w][ld] -> NCHW on the host
        std::vector<float> tmp((size_t)n * s.h * s.w * s.ld);
        PB_HIP(hipMemcpy(tmp.data(), s.ptr, tmp.size() * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < n; ++b)
            for (int64_t c = 0; c < s.c; ++c)
                for (int64_t p = 0; p < s.h * s.w; ++p) out[((int64_t)b * s.c + c) * s.h * s.w + p] = tmp[((int64_t)b * s.h * s.w + p) * s.ld + c];
        return total;
    }
    float *tmp = nullptr;
    PB_HIP(hipMalloc((void **)&tmp, total * 4));
    int r = launch_nhwc_f16_to_nchw_f32(stream, (const f16 *)s.ptr, tmp, n, (int)s.c, (int)s.h, (int)s.w, (int)s.ld);
    if (r) return r;
    PB_HIP(hipStreamSynchronize(stream));
    PB_HIP(hipMemcpy(out, tmp, total * 4, hipMemcpyDeviceToHost));
    if (s.bstride) {             // split map: add the residual parts (bstride = their half offset)
        std::vector<float> lo_part((size_t)total);
        r = launch_nhwc_f16_to_nchw_f32(stream, (const f16 *)s.ptr + s.bstride, tmp, n, (int)s.c, (int)s.h, (int)s.w, (int)s.ld);
        if (r) return r;
        PB_HIP(hipStreamSynchronize(stream));
        PB_HIP(hipMemcpy(lo_part.data(), tmp, total * 4, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < total; ++i) out[i] += lo_part[(size_t)i];
    }
    PB_HIP(hipFree(tmp));
    return total;
}
