// mask_mmdet band engine (SOLOv2: ResNet + FPN + SOLOV2Head + Matrix NMS + the band's mask accumulation).
#pragma once
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "engine_base.h"
#include "mask_kernels.h"

class MaskEngine : public EngineBase {
  public:
    MaskEngine(int device, const pb_mask_cfg &cfg) : EngineBase(device), cfg_(cfg) {}
    ~MaskEngine() override;
    int load(const pb_tensor *w, int n);
    // frames: device uint8 [n, H, W, 3] RGB.  mask_out: device uint8 [n, H, W, 3] (the band's "mask ids" image).
    int infer(const uint8_t *frames, int n, int H, int W, float confidence, const int32_t *keep, int n_keep, uint8_t *mask_out);
    int64_t get_stage(const char *name, float *out, int64_t cap, int64_t shape[4]);
    // --sdf of the band (reference mask_mmdet.py:64-69,150-152): with tables set (n_tab = ncap + 1 bytes each; 0 turns it off) every
    // infer() writes the clamped signed distance field of each frame's id image into its green channel before returning;
    // sdf_green applies it to id images already on the device (mask_kernels.hip sdf_*_kernel)
    int set_sdf(const uint8_t *tab_out, const uint8_t *tab_in, int n_tab);
    int sdf_green(uint8_t *masks, int n, int H, int W);
    static void net_size(const pb_mask_cfg &cfg, int H, int W, int *nh, int *nw, int *Hp, int *Wp);

    // results of the last infer(): per frame, score-descending (what format_results would hand to the band)
    struct Instances {
        std::vector<float> scores;
        std::vector<int32_t> labels;
        std::vector<uint8_t> masks;      // [count, H, W] bytes of 0 / 1, filled only when `debug` is set
        int candidates = 0;              // grid cells over score_thr (before the area filter)
    };
    const std::vector<Instances> &results() const { return results_; }
    // hooks of ONE infer() call (the host-pointer pipeline of pb_mask_infer_batch, abi.hip): chunk_begin(first, m) is called before the first
    // launch of the chunk of frames [first, first + m) - it makes the ctx stream wait for that chunk's H2D copy - and chunk_end(first, m) after
    // the chunk's last launch has been enqueued (its id images then go back to the host while the next chunk runs).  Cleared by infer().
    std::function<int(int, int)> chunk_begin, chunk_end;
    int frames_per_chunk() const { return cfg_.max_batch > 0 ? cfg_.max_batch : 1; }
    bool sdf_on() const { return sdf_ncap_ > 0; }

  private:
    struct GN { float *g = nullptr, *b = nullptr; int C = 0; };
    struct ConvGN { PackedW w; GN gn; };
    struct Bneck { PackedW c1, c2, c3, ds; bool has_ds = false; int stride = 1, planes = 0, inpl = 0; };

    int prepare(int n, int H, int W);
    int run_chunk(const uint8_t *frames, int n, int first, float confidence, const std::vector<uint8_t> &keep_class,
                  uint8_t *mask_out);
    int backbone(int n);
    int neck(int n);
    int head(int n);
    int head_level(int n, int lvl);
    int post_chunk(int n, int first, float confidence, const std::vector<uint8_t> &keep_class, uint8_t *mask_out);
    int ensure_post(size_t cands, int frames);

    int conv_f32(const f16 *in, int cC, int cLd, int n, int H, int W, const PackedW &w, float *out, int ldo);
    // channel counts / strides below are those of ONE part; L() / lo() turn them into the row stride and residual offset of a split map
    int conv_gn_relu(const f16 *in, int cC, int cLd, int n, int H, int W, int k, const ConvGN &c, f16 *tmp, f16 *out, int ldo, int dup = 0);
    int L(int c) const { return c * (1 + sa_); }
    int lo(int c) const { return sa_ ? c : 0; }
    int load_gn(const std::string &name, int C, GN &out);
    int load_conv_gn(const std::string &name, ConvGN &out, int ci_pad = 0);

    pb_mask_cfg cfg_;
    int sa_ = 0;                 // PB_PREC_SPLIT: feature maps are [hi | lo] fp16 pairs (mask_engine.hip load)

    // weights
    PackedW stem_;
    std::vector<Bneck> blocks_[4];
    PackedW lat_[4], fpnc_[4];
    ConvGN mfc_[4][3], mfpred_;
    std::vector<ConvGN> kconv_, cconv_;
    PackedW conv_cls_, conv_kernel_;

    // plan
    int pB_ = 0, pH_ = 0, pW_ = 0;
    int nh_ = 0, nw_ = 0, Hp_ = 0, Wp_ = 0;
    int lh_[6] = {}, lw_[6] = {};            // feature sizes at strides 4, 8, 16, 32, 64 (index 0..4); [5] = stride 2
    int pts_ = 0, goff_[6] = {};
    int *xt_ = nullptr, *yt_ = nullptr;
    f16 *img_ = nullptr, *stem_out_ = nullptr, *pool_ = nullptr;      // img_: 4 x 4 space-to-depth of the padded input
    float *chw_ = nullptr;
    f16 *sx_[4][2] = {}, *st1_[4] = {}, *st2_[4] = {}, *sds_[4] = {};
    const f16 *c_[4] = {};
    f16 *latb_[4] = {}, *p_[5] = {};
    f16 *mt_[2] = {}, *mg_[2] = {}, *macc_ = nullptr, *mf_ = nullptr, *p5cc_ = nullptr;
    f16 *fcc_[5] = {}, *rs_[5] = {}, *grid_[5] = {}, *hk_[5][3] = {};
    float *gstl_[5] = {}, *gaffl_[5] = {};
    float *kp_ = nullptr, *cl_ = nullptr, *cs_ = nullptr;
    float *gst_ = nullptr, *gaff_ = nullptr;
    // streams: helpers launch on cur_ (main stream, or one of the per-level streams while the head is forked)
    hipStream_t ls_[5] = {};
    hipEvent_t ev_fork_ = nullptr, ev_join_[5] = {};
    float *gst_cur_ = nullptr, *gaff_cur_ = nullptr;

    // post-processing scratch (grown on demand, outside the arena)
    size_t post_cap_ = 0;
    int post_frames_ = 0;
    f16 *pk_ = nullptr;
    unsigned long long *bits_ = nullptr;
    float *plog_ = nullptr, *pstat_ = nullptr, *inter_ = nullptr, *sig_ = nullptr, *nmsf_ = nullptr;
    int *pidx_ = nullptr, *nmsi_ = nullptr;
    uint8_t *use_ = nullptr, *inst_ = nullptr;
    size_t inst_bytes_ = 0;
    std::vector<float> h_scores_;

    uint8_t *sdf_tab_ = nullptr, *sdf_g_ = nullptr;      // [tab_out | tab_in], [gm | gb | cnt] scratch
    int sdf_ncap_ = 0;
    size_t sdf_px_ = 0;

    std::vector<Instances> results_;
    int last_n_ = 0;
    std::map<std::string, Stage> stages_;
};
