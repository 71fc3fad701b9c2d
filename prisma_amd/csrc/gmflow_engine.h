// flow_gmflow band engine: GMFlow at the band's defaults (bands/flow_gmflow.py:223-255: feature_channels 128, 1 scale,
// 1 head, swin attention with 2 x 2 windows, global matching, global propagation, 6 transformer blocks, ffn x 4, padding_factor 16).
// Shares RaftEngine's frame prep, instance-norm encoder, convex upsampling, flow encode and consistency-mask kernels.
#pragma once
#include "gmflow_kernels.h"
#include "raft_engine.h"

class GmflowEngine : public RaftEngine {
  public:
    explicit GmflowEngine(int device) : RaftEngine(device) {}
    int load(const pb_tensor *w, int n) override;
    // same contract as RaftEngine::infer; `iters` is ignored (GMFlow is not iterative)
    int infer(const uint8_t *frames, int F, int H, int W, float scale, int iters, int backward, float *flow_out, uint8_t *rgb_out,
              float *maxdisp, uint8_t *mask_out = nullptr, float alpha1 = 0.05f, float alpha2 = 0.5f) override;
    // fp32 stages of the last call as [n, rows, cols]: "feat" [F, P, 128], "block0" / "tfeat" [2 pairs, P, 128] (token stream after the
    // first / last transformer block), "flow_match" / "flow_prop" [pairs * dirs, P, 2]
    int64_t get_stage(const char *name, float *out, int64_t cap, int64_t shape[4]) override;
    // --inference_size of the band (reference flow_gmflow.py:76-100): the network runs on a bilinear (align_corners) resize of the scaled frame to
    // (h, w) - multiples of 16, no padding - and the flow is resized back and rescaled; (0, 0) = off (InputPadder(16), the default)
    int set_inference_size(int h, int w);

  private:
    struct Layer {
        PackedW w1;                 // [q_s; k_s; v_s; k_c; v_c] x 128: every projection of the token stream as it enters the block
        PackedW merge_s, q_c, merge_c, mlp0, mlp2;
        float *ln1s_g, *ln1s_b, *ln1c_g, *ln1c_b, *ln2c_g, *ln2c_b;
    };
    int prepare_g(int F, int H, int W, float scale, int dirs);
    int upload(const std::string &name, int n, float **dst);
    int gemm32(const f16 *A, int lda, int64_t M, const PackedW &w, float *out, int ldo);
    int gemm16(const f16 *A, int lda, int64_t M, const PackedW &w, f16 *out, int ldo, int act, int lo_off);
    int attention(const Attn128Args &a, double keys_per_query);
    struct FStage { const float *ptr; int64_t n, rows, cols; };

    Layer layers_[6];
    PackedW ffq_, ffk_, up0_, up2_;
    GmGeom g_{};
    int gF_ = 0, gH_ = 0, gW_ = 0, gD_ = 0;
    int isz_h_ = 0, isz_w_ = 0;
    float *gupi_ = nullptr;           // flow at the inference size, before the resize back
    float gS_ = 0.f;
    int ldvP_ = 0;
    float *feat_ = nullptr, *pos_ = nullptr, *X_ = nullptr, *Y1_ = nullptr, *Yq_ = nullptr, *Ow_ = nullptr, *M_ = nullptr, *Om_ = nullptr,
          *flowm_ = nullptr, *flowp_ = nullptr, *gmask_ = nullptr, *gup_ = nullptr, *blk0_ = nullptr;
    f16 *Xs_ = nullptr, *Qw_ = nullptr, *Kw_ = nullptr, *Kcw_ = nullptr, *Vtw_ = nullptr, *Vtcw_ = nullptr, *Os_ = nullptr, *cat_ = nullptr,
        *Hs_ = nullptr, *gridvt_ = nullptr, *Vtf_ = nullptr, *qs_ = nullptr, *ks_ = nullptr, *umap_ = nullptr, *u1_ = nullptr;
    int8_t *region_ = nullptr;
    unsigned *gmaxd_ = nullptr;
    std::map<std::string, FStage> fstages_;
};
