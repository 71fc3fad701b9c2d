// Shared plumbing of the convolutional band engines (flow_raft, mask_mmdet): the named-weight map, fp16 GEMM packing with
// folded BatchNorm, a two-pass arena, per-family kernel timing, and the implicit-GEMM / dense GEMM launch helpers.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "engine.h"

class EngineBase {
  public:
    explicit EngineBase(int device) : device(device) {}
    virtual ~EngineBase();
    int stats(pb_kernel_stat *out, int cap);

    hipStream_t stream = nullptr;
    int device = 0;
    bool debug = false;
    KernelTimer timer;
    int conv_tile = TILE_AUTO;
    int split_w_ = 0;            // PB_PREC_SPLIT: weights packed as hi + lo fp16, two passes over K (set before load)
    int mx_ = 0;                 // ... and where activations are split too (sa) the residual parts are e4m3: maps [hi | hi8 | lo8],
                                 // weights [w_hi | w_lo8 | w_hi8] per tap, fp8 tiles through the MX-scaled MFMA (PackedW::mx3)
    // power-of-two storage scales of the e4m3 copies.  hi8 = e4m3(x 2^kLo8Pa), lo8 = e4m3((x - fp16(x)) 2^(kLo8Pa + 12)); e4m3 saturates at
    // 448, so the copy of |x| > 448 / 2^kLo8Pa clamps and that element's correction term degrades to the single-pass error.  Round 3:
    // 2^0 (saturation at 448; was 2^3 = 56) - a correction term needs a few significant bits of a NORMAL e4m3 (|x| >= 2^-6) and what
    // lies below contributes |x| 2^-12 |w| at most (tools/precision_budget.py --scales; tests/test_gpu_outliers.py)
    const int kLo8Pa = pb_env_int("PB_LO8_POW", 0);
    int pack_mx2_ = 0;           // while set, pack() lays weights out as [w_hi fp16 | w_lo e4m3] per tap (PackedW::mx2): the layer's input
                                 // map carries an fp8 copy after its fp16 part ([a16 (Ctot) | a8 (Ctot bytes)], scaled by 2^kMx2Pa)
    const int kMx2Pa = pb_env_int("PB_A8_POW", 0);      // [a16 | a8] maps: a8 = e4m3(a 2^kMx2Pa); 2^0 since round 3 (was 2^4: |a| > 28 clamped)
    int pack_tapin_ = 0;         // while set, pack() stores convolution weights (taps > 1) in slice-major K order (gemm.h cTapInner)
    const f16 *zero_page() const { return zero_; }

  protected:
    // kernel families of the per-launch timer; convolutions are split by the GEMM kernel that runs them (launch_gemm's choice)
    enum { F_GEMM = 0, F_CONV = 1, F_ATTN = 2, F_LN = 3, F_ELT = 4, F_PP = 5, F_CONV128 = 6, F_CONV64 = 7, F_COUNT = 8 };

    // create the ctx stream, index the float32 tensors by name, allocate the zero page
    int begin_load(const pb_tensor *w, int n);
    const pb_tensor *find(const std::string &name) const;
    // src: host fp32 [N, K] in GEMM order -> device fp16 [round_up(N, 256), Kpad] (+ fp32 bias, zero padded)
    // taps: src is [N][taps][K / taps] (only matters with split_w_: the hi / lo segments alternate per tap)
    // sa: the activation operand of this weight is a split-fp16 map [hi | lo] (per tap [w_hi | w_hi | w_lo]; only with split_w_)
    int pack(const float *src, int N, int K, int Kpad, PackedW &out, const float *bias, int taps = 1, int sa = 0);
    // eval-mode BatchNorm2d (eps 1e-5) as a per-channel (scale, shift)
    int fold_bn(const std::string &bn, int C, std::vector<float> &scale, std::vector<float> &shift);
    // conv weight [co, ci, kh, kw] (+ bias) -> rows [co][(ky * kw + kx) * round_up(ci, 64) + c], optional per-output affine
    // ci_pad: input channels the rows are padded to (default round_up(ci, 64))
    int pack_conv(const std::string &name, bool has_bias, const float *scale, const float *shift, PackedW &out, int sa = 0, int ci_pad = 0);

    // direct launch_gemm callers: K, and with split-fp16 weights the K wrap / channel bookkeeping of gemm.h (a.cC, a.cLd of
    // ONE part must be set before the call for convolutions)
    void set_weights(GemmArgs &a, const PackedW &w, bool is_conv) const;

    void *carve(size_t bytes);
    // grow the arena to the planned size (after the planning pass) and zero it
    int commit_arena(const char *what);

    // helpers launch on cur_ (the ctx stream unless a derived engine forks work onto side streams)
    void tic(int fam, double flops, double bytes, double passes = 1.0);
    void toc();
    // fusion hooks of one conv() call: a second (ReLU'd) copy of the output, and the SepConvGRU epilogues (gemm.h ACT_GRU_*)
    struct ConvFuse { f16 *out2 = nullptr; float *gru_h = nullptr; const f16 *gru_z = nullptr; f16 *gru_rh = nullptr; int gru_ld = 384;
                      const f16 *add2 = nullptr; };       // add2: a second skip tensor (same row stride as the output), added next to add1
    // lo_off != 0: out (and add1) are split-fp16 maps, the rounding residual of every output goes to +lo_off (gemm.h)
    int conv(const f16 *in, int cC, int cLd, int n, int H, int W, int kh, int kw, int stride, const PackedW &w, f16 *out, int ldo,
             int act, int pre_relu = 0, const f16 *add1 = nullptr, const ConvFuse *fuse = nullptr, int lo_off = 0, int a8_rel = 0,
             int o8_off = 0);
    // a8_rel (mx2 weights): half-offset from `in` to the fp8 copy of its channels (0 = right after the cC channels); o8_off: byte
    // offset, from the output row, of the fp8 copy the epilogue also stores (0 = none)
    int dense(const f16 *A, int lda, int64_t M, const PackedW &w, f16 *out, int ldo, int act, const f16 *add1 = nullptr, int o8_off = 0,
              int a_pa = -1, int lo_off = 0);          // a_pa: power-of-two scale the A operand's fp8 copy was stored with (default kMx2Pa)

    std::map<std::string, const pb_tensor *> tmap_;
    std::vector<void *> owned_;                 // permanent device allocations (weights), freed by the destructor
    f16 *zero_ = nullptr;
    char *arena_ = nullptr;
    size_t arena_bytes_ = 0, arena_off_ = 0;
    bool planning_ = false;
    hipStream_t cur_ = nullptr;
    std::vector<size_t> open_;                  // tic / toc nesting across streams
};
