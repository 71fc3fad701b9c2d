// flow_raft band engine: weight packing (BatchNorm folded), arena planning, launch sequence.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "engine_base.h"
#include "raft_kernels.h"

class RaftEngine : public EngineBase {
  public:
    explicit RaftEngine(int device) : EngineBase(device) {}
    ~RaftEngine() override { for (auto &kv : snaps_) if (kv.second) hipFree(kv.second); }
    virtual int load(const pb_tensor *w, int n);
    // frames: device uint8 [F, H, W, 3].  Outputs are device pointers (any may be null):
    //   flow_out [F-1, dirs, sh, sw, 2] fp32, rgb_out [F-1, dirs, sh, sw, 3] u8, maxdisp [F-1, dirs]
    virtual int infer(const uint8_t *frames, int F, int H, int W, float scale, int iters, int backward, float *flow_out,
                      uint8_t *rgb_out, float *maxdisp, uint8_t *mask_out = nullptr, float alpha1 = 0.05f,
                      float alpha2 = 0.5f);
    virtual int64_t get_stage(const char *name, float *out, int64_t cap, int64_t shape[4]);
    static void out_size(int H, int W, float scale, int *sh, int *sw);

  protected:            // shared with GmflowEngine (gmflow_engine.h): GMFlow's CNNEncoder is this encoder without conv biases
    struct Enc {                    // BasicEncoder weights (BN folded for cnet)
        PackedW stem, l[3][2][2], ds[3], out;
    };
    int pack_encoder(const std::string &en, bool bnf, bool conv_bias, Enc &E);
    int run_encoder(const Enc &E, bool inorm, int F, const f16 **x_out);
    void geometry(int H, int W, float scale, int factor);
    void carve_encoder(int F);
    int upload_resize_tables(int H, int W, float scale);
    int prepare(int F, int H, int W, float scale, int dirs);
    Enc fnet_, cnet_;
    PackedW convc1_, convc2_, convf1_, convf2_, convm_, zr_[2], q_[2], fh1_, fh2_, mk0_, mk2_;
    // convf1 as a direct kernel on the fp32 flow field (raft_kernels.hip convf1_kernel; PB_CONVF1_DIRECT=0: im2col + GEMM as in rounds 1-3)
    f16 *f1w_ = nullptr;
    int f1_passes_ = 1;
    bool f1_direct_ = true;
    // context hoist (raft_engine.hip load()): zr_ / q_ then cover [h | motion] only, zr_in_ / q_in_ the context features' 128 channels
    PackedW zr_in_[2], q_in_[2];
    int hoist_ = 0;
    f16 *gz_[2] = {}, *gq_[2] = {};             // hoisted shares: [hi plane | lo plane], rows x 256 (z | r) and rows x Lhx (q)

    // plan
    int pF_ = 0, pH_ = 0, pW_ = 0, pD_ = 0;
    float pS_ = 0.f;
    int sh_ = 0, sw_ = 0, Hp_ = 0, Wp_ = 0, padl_ = 0, padt_ = 0, h8_ = 0, w8_ = 0, P_ = 0, P8_ = 0;
    int lh_[4] = {0, 0, 0, 0}, lw_[4] = {0, 0, 0, 0};
    int *xi_ = nullptr, *xc_ = nullptr, *yi_ = nullptr, *yc_ = nullptr;
    f16 *img_ = nullptr;                              // padded frames, 4 x 4 space-to-depth: [F, Hp / 4, Wp / 4, 64] fp16
    f16 *r1_[7] = {}, *r2_[7] = {}, *r3_[7] = {};     // scratch maps at 1/2, 1/4, 1/8 resolution: t1 t2 t3 outA outB stem stem_n
    float *st_[3] = {};                               // instance-norm statistics {mean, rstd} per (frame, channel)
    float *stp_ = nullptr;                            // ... and their per-chunk partial sums
    f16 *fmap_ = nullptr, *ctx_ = nullptr;
    f16 *pyr_[4] = {};                                // correlation volume levels: fp16 [pairs * P, pld_] in 8 x 8 target tiles
    f16 *ftile_[4] = {};                              // target features / 16 in tile order (B operand of the volume GEMM)
    int lwp_[4] = {};
    f16 *fpool_[4] = {};                              // avg-pooled target features of levels 1..3
    int pld_[4] = {};
    float *h32_ = nullptr, *flow_ = nullptr, *mask_ = nullptr, *up_ = nullptr;
    f16 *hx_ = nullptr, *hx2_ = nullptr, *corr_ = nullptr, *c1_ = nullptr, *corflo_ = nullptr, *fa_ = nullptr, *f1_ = nullptr,
        *zrb_ = nullptr, *fh_ = nullptr, *m0_ = nullptr;
    unsigned *maxd_ = nullptr;
    int last_nd_ = 0;
    int upd8_ = 0;               // the update block's maps carry fp8 copies and its weights e4m3 residuals (MX segments; raft_engine.hip load)
    std::map<std::string, Stage> stages_;
    // debug snapshots (pb_set_profiling bit 2): copies of buffers later iterations overwrite - the initial hidden state, the first
    // lookup's output and the flow after the first iteration, the intermediates the reference goldens hold (tests/golden/raft_*.npz)
    std::map<std::string, void *> snaps_;
    int snapshot(const char *name, const void *src, size_t bytes, const Stage &as);
};
