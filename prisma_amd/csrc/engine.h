// Depth-Anything band engine: weight packing, arena planning and the per-batch launch sequence.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/prisma_bands.h"
#include "common.h"
#include "gemm.h"
#include "kernels.h"
#include "zoe_kernels.h"

// The ctx stream of a band engine.  `mask_env` names an environment variable with a CU mask as comma-separated 32-bit hex words (bit i of the
// concatenation = CU i of the queue's mask; on gfx950 the driver deals the bits round-robin over the 8 XCDs, so 0f0f0f0f,... keeps XCDs 0-3):
// an experiment switch for running two bands on disjoint parts of the chip (tools/overlap_bench.py --cu-masks).  Unset = the whole device.
static inline hipError_t pb_create_stream(hipStream_t *s, const char *mask_env) {
    const char *e = mask_env ? getenv(mask_env) : nullptr;
    if (e && *e) {
        uint32_t words[8];
        int n = 0;
        for (const char *p = e; *p && n < 8;) {
            char *end = nullptr;
            words[n++] = (uint32_t)strtoul(p, &end, 16);
            if (!end || end == p) break;
            p = *end == ',' ? end + 1 : end;
        }
        if (n > 0) return hipExtStreamCreateWithCUMask(s, (uint32_t)n, words);
    }
    // <mask_env>_PRIO = -1 / 1: a higher- / lower-priority queue (hipStreamCreateWithPriority; 0 or unset = default).  Experiment switch: with two
    // bands at once the band with the large persistent GEMMs gets more than half of the chip (tools/overlap_bench.py 2way-fprio / 2way-dprio)
    if (mask_env) {
        const std::string pe = std::string(mask_env) + "_PRIO";
        const char *p = getenv(pe.c_str());
        if (p && *p && atoi(p) != 0) {
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) return hipStreamCreateWithPriority(s, hipStreamNonBlocking, atoi(p) < 0 ? hi : lo);
        }
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

struct PackedW {
    f16 *w = nullptr;        // [Npad, K] fp16, K % 64 == 0
    float *bias = nullptr;   // [N] fp32 or null
    int N = 0, K = 0, Kreal = 0;
    int Nreal = 0;           // output columns that are the layer's own when N holds padding rows (RAFT's motion conv: 126 of 128); 0 = N.  Only the FLOP / byte counters read it
    // split-fp16 packing (gemm.h kwrap): per tap the K axis holds the segments [w_hi | w_hi (if sa) | w_lo (if sw)] of Cseg
    // channels each; the activation operand must then be [hi | lo (if sa)] with Cseg channels per part
    int sa = 0, sw = 0, Cseg = 0;
    // MX-fp8 correction (gemm.h nk16): rows are [w_hi fp16 (Kin) | w_lo e4m3 (Kin bytes, scaled by 2^mx_pw)]; K counts 64-half units of the row
    int nk16 = 0, mx_pw = 0;
    // mx3: weights AND activations split with e4m3 residual parts: per tap [w_hi fp16 (Cseg) | w_lo8 (Cseg bytes) | w_hi8 (Cseg bytes)]
    // = 2 Cseg halfs, meeting a split map's [hi | hi8 | lo8] pixel (gemm.h mx_period, lo8); K = taps x 2 Cseg
    int mx3 = 0;
    int mx2 = 0;             // weights-only split with the residual as e4m3: per tap [w_hi fp16 (Cseg) | w_lo8 (Cseg bytes)] = 1.5 Cseg halfs
    // conv weights in slice-major K order (gemm.h cTapInner): the (taps x K / (64 taps)) grid of 128-byte blocks of every row, transposed
    int tapin = 0, taps = 1;
    // packed-channel copy (gemm.h Wcw, conv_walk.h conv_cw3_word): mx3 convolution weights whose Cseg input channels hold cwC < Cseg real ones, with a K axis
    // that lists only those - read by the 128 x 96 tile's chunk walk (EngineBase::pack_conv builds it for 64 < N <= 96)
    f16 *wcw = nullptr;
    int Kcw = 0, nk16cw = 0, cwC = 0;
};
unsigned char pb_f32_to_e4m3(float x);     // OCP e4m3fn, round to nearest even, saturating (engine.hip)
// slice-major K order of packed conv weights (gemm.h cTapInner): every row of `rowlen` halfs is a (taps x S) grid of 128-byte blocks; transposed in place
void pb_rows_slice_major(f16 *rows, int64_t nrows, int64_t rowlen, int taps);

struct Stage {
    const void *ptr;
    int kind;                // 0: fp32 rows [n, rows, cols] with batch stride; 1: NHWC fp16 map; 2: fp32 map [n,h,w]; 3: fp16 rows
    int64_t n, c, h, w, ld, bstride;
};

struct KernelTimer {
    struct Rec { int fam; hipEvent_t a, b; double flops, bytes, exec; const char *name; };   // name: GEMM launches, the kernel symbol (gemm.h)
    // per family (non-GEMM kernels) / per GEMM kernel symbol sums of the records
    int collect(const char *const *fam_names, int nfam, pb_kernel_stat *out, int cap);
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    bool enabled = false;
    bool accumulate = false;      // keep the records of earlier infer calls (one query after a timed loop)
    hipEvent_t get();
    void reset() { if (accumulate) return; recs.clear(); used = 0; }
    void clear() { recs.clear(); used = 0; }
    ~KernelTimer();
};

class DepthEngine {
  public:
    DepthEngine(int device, const pb_depth_cfg &cfg);
    ~DepthEngine();
    int load(const pb_tensor *w, int n);
    int infer(const uint8_t *frames_dev, int n, int H, int W, float *depth_out, uint8_t *rgb_out, float *mn,
              float *mx, int flip);
    int64_t get_stage(const char *name, float *out, int64_t cap, int64_t shape[4]);
    int stats(pb_kernel_stat *out, int cap);

    hipStream_t stream = nullptr;
    int device = 0;
    bool debug = false;
    int gemm_tile = TILE_AUTO, conv_tile = TILE_AUTO;
    KernelTimer timer;

    // scratch for op-level tests
    int dev_alloc(void **p, size_t bytes);
    const f16 *zero_page() const { return zero_; }
    int max_batch() const { return cfg_.max_batch > 0 ? cfg_.max_batch : 1; }

  private:
    int prepare(int B, int H, int W);
    int run_chunk(const uint8_t *frames, int n, float *depth_out, uint8_t *rgb_out, float *mn, float *mx, int flip);
    int vit(int n);
    int head(int f0, int n);
    int tail(int n, float *depth_out, uint8_t *rgb_out, float *mn, float *mx, int flip);
    // ZoeDepth metric head (engine_zoe.hip)
    int load_metric();
    int plan_metric(int B, int H, int W);
    int metric_tables(int H, int W);
    int metric_head(int n);
    int gemm(int amode, int epi, GemmArgs &a, const PackedW &w, int tile = TILE_AUTO);
    int conv3(const f16 *in, int inC, int n, int H, int W, const PackedW &w, f16 *out, f16 *out2, const f16 *add1,
              const f16 *add2, int act, int stride, int outC);
    void *carve(size_t bytes);
    int upload_f32(const float *src, size_t n, float **dst);
    // taps > 1: src is [N][taps][K / taps] and every tap is padded to Kpad / taps; sa / sw: split-fp16 segments (PackedW)
    int pack(const float *src, int N, int K, int Kpad, PackedW &out, const float *bias, int taps = 1, int sa = 0, int sw = 0);
    // dense weights with the split residual as an MX-fp8 segment (K % 128 == 0): PackedW::nk16
    int pack_mx(const float *src, int N, int K, PackedW &out, const float *bias);
    const pb_tensor *find(const std::string &name) const;
    void tic(int fam, double flops, double bytes, double passes = 1.0);
    void toc();
    void snapshot(const std::string &name);

    pb_depth_cfg cfg_;
    // PB_PREC_SPLIT (tools/precision_budget.py): ViT linears and the metric head keep their weights as hi + lo (2 passes);
    // the DPT head keeps weights AND feature maps as hi + lo (3 passes; maps are [hi | lo] per pixel, hs_ = 2)
    int head_tapin_ = 1;         // the head's 3x3 convolutions walk K slice-major (gemm.h cTapInner; PB_TAPIN=0 turns it off)
    int vit_sw_ = 0, head_sa_ = 0, head_sw_ = 0, hs_ = 1;
    // vit_mx_: the ViT linears' weight residual runs as an MX-fp8 segment (half the matrix-pipe time of an fp16 pass); their A
    // operands (LayerNorm out, attention out, GELU out) then carry an fp8 copy after the fp16 part of each row (row stride 1.5 K)
    int vit_res_ = 15;           // which ViT linears keep their weight residual: bit 0 qkv, 1 proj, 2 fc1, 3 fc2 (engine.hip load())
    int vit_mx_ = 0, head_mx_ = 0;              // head_mx_: the DPT head's maps / weights carry e4m3 residual parts (PackedW::mx3)
    // storage scales of the e4m3 copies (engine_base.h kLo8Pa has the reasoning): head maps hi8 = e4m3(x 2^kLo8Pa), lo8 = e4m3(lo 2^(kLo8Pa + 12));
    // ViT token rows a8 = e4m3(a 2^kMxPa).  2^0 since round 3: the copies saturate at |x| = 448 (were 2^3 / 2^4: 56 / 28)
    const int kLo8Pa = pb_env_int("PB_LO8_POW", 0);
    const int kMxPa = pb_env_int("PB_A8_POW", 0);
    int batch_cap(int H, int W) const;          // frames per chunk the 32-bit tensor offsets allow for this frame size
    std::map<std::string, const pb_tensor *> tmap_;
    std::vector<void *> owned_;                 // permanent device allocations (weights)
    // weights
    PackedW patch_;
    float *cls_ = nullptr;
    std::vector<float> pos_host_;               // [1 + g*g, D]
    struct Block { float *ln1g, *ln1b, *ln2g, *ln2b, *ls1, *ls2; PackedW qkv, proj, fc1, fc2; };
    std::vector<Block> blocks_;
    float *normg_ = nullptr, *normb_ = nullptr;
    PackedW proj_[4], rs0_, rs1_, rs3_, rn_[4], outc_[4], rcu_[4][2][2], oc1_, oc2_;
    // output_conv2's 3 x 3 weights as a 1 x 1 GEMM [9 x 32, F / 2] over output_conv1's LOW-resolution map: row t * 32 + c = tap t of output channel c
    // (the convolution commutes with the bilinear resize in front of it; elementwise.hip dpt_tail_kernel sums the resized tap products).  PB_HEAD_TAIL=0:
    // the round-1..5 formulation (resize to the network size, implicit GEMM with the fused 1 x 1 there); the metric head keeps that one (it needs the
    // 32-channel activation at the network resolution)
    PackedW wz_;
    int head_tail_ = 1;
    f16 *z_ = nullptr;
    float *w2_ = nullptr;
    float b2_ = 0.f;
    f16 *zero_ = nullptr;
    // split-K workspace (gemm.h splitk): lent to every launch of a context created with max_batch = 1 - one frame per call is the latency
    // configuration (BASELINE.json configs[1]), its residual GEMMs and low-resolution head convolutions have 20-160 tiles for 256 CUs.
    // Contexts with max_batch > 1 never split, so a frame's bits do not depend on the size of the batch it arrives in; they do differ (within
    // the parity tolerance) between a max_batch = 1 context and the others.  PB_SPLITK=0 turns it off, PB_SPLITK_ALWAYS=1 lends it to every ctx.
    float *sk_ws_ = nullptr;
    static constexpr int64_t kSkFloats = (int64_t)512 * 128 * 128;      // 512 slices of a 128 x 128 fp32 tile: 32 MB

    // plan
    int pB_ = 0, pH_ = 0, pW_ = 0, last_n_ = 0;
    int hB_ = 1;                                // frames per DPT-head chunk (<= batch_cap: 32-bit offsets of the high-resolution split maps)
    int nh_ = 0, nw_ = 0, gh_ = 0, gw_ = 0, ntok_ = 0, ntp_ = 0, P_ = 0;
    int lh_[4] = {0, 0, 0, 0}, lw_[4] = {0, 0, 0, 0};      // DPT level sizes (level 0 = finest)
    char *arena_ = nullptr;
    size_t arena_bytes_ = 0, arena_off_ = 0;
    bool planning_ = false;
    // arena buffers
    int *xi_ = nullptr, *yi_ = nullptr;
    float *xw_ = nullptr, *yw_ = nullptr, *pos_ = nullptr;
    f16 *patchA_ = nullptr, *Y_ = nullptr, *Q_ = nullptr, *K_ = nullptr, *Vt_ = nullptr, *AO_ = nullptr, *Hd_ = nullptr;
    float *X_ = nullptr;
    f16 *feat_[4] = {nullptr, nullptr, nullptr, nullptr};
    f16 *pj_[4] = {}, *lay_[4] = {}, *rnraw_[4] = {}, *rnrelu_[4] = {}, *tmp_[4] = {}, *sraw_[4] = {}, *srelu_[4] = {},
        *yb_[4] = {}, *ocb_[4] = {}, *path_[4] = {};
    f16 *o1_ = nullptr, *up_ = nullptr;
    float *netd_ = nullptr, *full_ = nullptr;
    unsigned *mm_ = nullptr;
    std::map<std::string, Stage> stages_;
    std::map<std::string, float *> snaps_;
    // metric head: weights, buffers (arena), Pillow resize tables
    struct Mlp2 { PackedW a, b; };
    PackedW zconv2_;
    Mlp2 zseed_, zsproj_, zproj_[4], zattr_[4], zclb_;
    f16 *act32_ = nullptr, *zx0_ = nullptr, *zh_ = nullptr, *zemb_[2] = {}, *zxa_ = nullptr, *zcat_ = nullptr, *zmid_ = nullptr;
    float *zA_ = nullptr, *zbins_[2] = {}, *zpt_ = nullptr, *md_ = nullptr, *ptmp_ = nullptr;
    int *pxb_ = nullptr, *pyb_ = nullptr;
    double *pxk_ = nullptr, *pyk_ = nullptr;
    int pxks_ = 0, pyks_ = 0;
};
