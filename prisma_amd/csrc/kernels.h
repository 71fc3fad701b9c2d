// Launchers of the non-GEMM kernels (elementwise.hip, attention.hip).
#pragma once
#include "common.h"

// o8_off != 0: also store fp8(o * o8_scale) at byte offset o8_off of every output row (gemm.h nk16)
int launch_attention(hipStream_t s, const f16 *q, const f16 *k, const f16 *vt, f16 *o, int B, int heads, int ntp,
                     int ntok, int ldo, int variant = 0, int o8_off = 0, float o8_scale = 16.f);

// LayerNorm(eps) of fp32 rows -> fp16 rows.  Input row r = (b, t) of [B, ntp, D]; only t < ntok are
// normalised.  drop_cls = 0: output row = input row (same [B, ntp] indexing, ld = D).
// drop_cls = 1: output is compact [B, ntok-1, D] without the class token (DPT taps).
// ldy = output row stride (0 = D); lo_off != 0: also store the rounding residual of every output at +lo_off (split fp16).
int launch_layernorm(hipStream_t s, const float *x, const float *g, const float *b, f16 *y, int B, int ntp, int ntok,
                     int D, float eps, int drop_cls, int ldy = 0, int lo_off = 0, int o8_off = 0, float o8_scale = 16.f, int lo8_pa = -1);

// resid[b, 0, :] = cls + pos[0]
int launch_cls_rows(hipStream_t s, float *resid, const float *cls, const float *pos, int B, int ntp, int D);

// uint8 RGB frames -> bicubic resize -> normalise -> patch-major fp16 matrix [B * gh*gw, Kp]
// (k = c*196 + py*14 + px, zero padded to Kp).  Tap tables: idx [n,4] int, w [n,4] float.
int launch_preprocess(hipStream_t s, const uint8_t *frames, int B, int H, int W, int nh, int nw, const int *xi,
                      const float *xw, const int *yi, const float *yw, f16 *out, int Kp, float *chw_out);

// NHWC fp16 bilinear resize (torch F.interpolate semantics), C % 8 == 0 with channel stride ldc.
// lo_off != 0: x and y are split-fp16 maps [hi | lo] (lo at element offset +lo_off inside a pixel).
int launch_bilinear_nhwc(hipStream_t s, const f16 *x, f16 *y, int B, int H, int W, int OH, int OW, int C, int ldc,
                         int align_corners, int lo_off = 0, int lo8_pa = -1);

// DPT head tail from the low-resolution tap-product map z (elementwise.hip dpt_tail_kernel): bilinear(align_corners=True) resize to [OH, OW], sum of the nine
// taps with the 3 x 3 convolution's zero padding, + bias -> ReLU -> 1 x 1 (w2, b2) -> ReLU -> out [B, OH, OW] fp32.  z pixel layouts as launch_bilinear_nhwc's.
int launch_dpt_tail(hipStream_t s, const f16 *z, int B, int H, int W, int ldz, int lo_off, int lo8_pa, const float *bias, const float *w2, float b2,
                    float *out, int OH, int OW);

// net depth [B, nh, nw] fp32 -> bilinear(align_corners=False) -> [B, H, W] fp32 (optional) and
// per-frame min/max (ordered-uint atomics in mm[2*B]); then heat encode to uint8 RGB.
int launch_depth_resize_minmax(hipStream_t s, const float *net, int B, int nh, int nw, float *out, int H, int W,
                               unsigned *mm);
int launch_minmax_only(hipStream_t s, const float *x, int B, int64_t per, unsigned *mm);
int launch_heat_encode(hipStream_t s, const float *depth, int B, int H, int W, const unsigned *mm, int flip,
                       uint8_t *rgb, float *mn, float *mx);
int launch_init_minmax(hipStream_t s, unsigned *mm, int B);
// write_depth(heatmap=True) of one float depth map (bands/common/io.py:138-172): heat ramp, Sobel-edge saturation, min / max packed in
// pixels (0, 0), (0, 1).  Scratch: mm [2] ordered-uint min / max, g2max [1], q [H * W] bytes; mnmx [2] floats out (optional).
int launch_still_encode(hipStream_t s, const float *depth, int H, int W, unsigned *mm, unsigned *g2max, uint8_t *q, int flip, int encode_range,
                        uint8_t *rgb, float *mnmx);

// layout converters used by the op-level tests and the stage dumps
int launch_nchw_f32_to_nhwc_f16(hipStream_t s, const float *x, f16 *y, int B, int C, int H, int W, int ldc, int relu);
int launch_nhwc_f16_to_nchw_f32(hipStream_t s, const f16 *x, float *y, int B, int C, int H, int W, int ldc);
int launch_f32_to_f16(hipStream_t s, const float *x, f16 *y, int64_t rows, int cols, int ld_out);
// attention128.hip: single-head attention over 128-wide heads (GMFlow's transformer / matching / propagation); Q, K [B, L, 128] fp16,
// Vt [B, 128, ldv] fp16, region [B, L] int8 or null (shifted-window mask), O [B, L, 128] fp32
int launch_attention128(hipStream_t s, const f16 *Q, const f16 *K, const f16 *Vt, const int8_t *region, float *O, int B, int L, int ldv);
// general form.  split = 1: Q, K rows are [hi (128) | lo (128)], Vt holds hi rows then lo rows, P is split in registers (three MFMA
// passes for S and for O).  vcols = 128, or 32 for a V padded to 32 columns (coordinates / flow); O is [B, L, vcols] fp32.
// Batch element b reads its queries at Q + b q_bstride and its keys / values at index b ^ kxor (K + (b ^ kxor) k_bstride, Vt + (b ^ kxor)
// v_bstride); strides in halfs, 0 = contiguous [B, L, ...] / [B, (2,) vcols, ldv]; v_shared: one Vt for every b; region [nreg, L]: b uses row b % nreg.
struct Attn128Args {
    const f16 *Q = nullptr, *K = nullptr, *Vt = nullptr;
    const int8_t *region = nullptr;
    int nreg = 0;
    float *O = nullptr;
    int B = 0, L = 0, ldv = 0, split = 0, vcols = 128, v_shared = 0, kxor = 0;
    int pv_single = 0;       // with split: P and V stay single fp16 (one MFMA pass for P V; Vt's hi rows only are read)
    int ldq = 0;             // row stride of Q and K in halfs (0: 256 with split, else 128); a non-split call may read the hi part of split rows
    int64_t q_bstride = 0, k_bstride = 0, v_bstride = 0;
};
int launch_attention128x(hipStream_t s, const Attn128Args &a);
int launch_f16_to_f32(hipStream_t s, const f16 *x, float *y, int64_t rows, int cols, int ld_in);
int launch_fill_random_f16(hipStream_t s, f16 *x, int64_t n, unsigned seed, float scale);
