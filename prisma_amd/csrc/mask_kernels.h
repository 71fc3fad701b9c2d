// Launchers of the mask_mmdet band's non-GEMM kernels (mask_kernels.hip).
#pragma once
#include "common.h"

// xt / yt: per destination column / row {i0, i1, c0, c1} (OpenCV INTER_LINEAR 11-bit tables); chw (optional): the
// normalised fp32 network input [n][3][Hp][Wp] for parity dumps.
int launch_mask_prep(hipStream_t s, const uint8_t *frames, int n, int H, int W, int nh, int nw, int Hp, int Wp, const int *xt,
                     const int *yt, f16 *out, float *chw, int split = 0);
// split != 0 / lo_* != 0: split maps, [hi (C) | lo (C)] fp16 pairs per pixel (mask_kernels.hip ld8 / st8); lo_* = half offset of the residual part
int launch_maxpool3x3s2(hipStream_t s, const f16 *x, f16 *y, int n, int H, int W, int C, int split = 0);
int launch_nearest_add(hipStream_t s, f16 *dst, const f16 *src, int n, int h, int w, int sh, int sw, int C, int split = 0);
int launch_subsample2(hipStream_t s, const f16 *x, f16 *y, int n, int H, int W, int C);
int launch_coord_concat(hipStream_t s, const f16 *x, f16 *y, int n, int h, int w, int C, int ldi, int lo_in = 0);
int launch_bilinear(hipStream_t s, const f16 *x, f16 *y, int n, int H, int W, int OH, int OW, int C, int ldi, int ldo,
                    int accumulate, int lo_in = 0, int lo_out = 0);
// `stats` is the partial-sum scratch: n * gn_chunks(HW) * C * 2 floats
enum { GN_CHUNK = 256 };
int gn_chunks(int HW);
int launch_gn_relu(hipStream_t s, const f16 *x, f16 *y, int n, int HW, int C, int ldc, int ldo, int groups, const float *gamma,
                   const float *beta, float *stats, float *aff, int lo_in = 0, int lo_out = 0, int dup = 0);
int launch_cls_points_nms(hipStream_t s, const float *logit, float *score, int n, int pts_total, int off, int g, int C);
int launch_gather_rows_f16(hipStream_t s, const float *src, const int *idx, f16 *dst, int count, int rows_pad, int cols, int split = 0);
int launch_mask_stats(hipStream_t s, const float *logit, int rows, int HW, int64_t ld, float thr, float *out);
int launch_bitpack_rows(hipStream_t s, const float *logit, int64_t ld, const int *idx, int count, int HW, float thr,
                        unsigned long long *bits);
int launch_mask_intersections(hipStream_t s, const unsigned long long *bits, int n, int words, float *inter, int ld);
int launch_matrix_nms(hipStream_t s, const float *inter, int ld, const float *area, const int *label, const float *score, int n,
                      float sigma, float *comp, float *out);
int launch_sigmoid_rows(hipStream_t s, const float *logit, int64_t ld, const int *idx, int count, int HW, float *sig);
int launch_band_accumulate(hipStream_t s, const float *sig, int k, int fh, int fw, int h, int w, int H, int W, float thr,
                           const uint8_t *use, uint8_t *out, uint8_t *inst);
// signed distance field of the id image into its green channel (mask_mmdet.py:64-69,150-152): gm / gb = n * H * W bytes of scratch each,
// cnt = n ints, tab_out / tab_in = ncap + 1 bytes each (the byte for squared distance i outside / inside the mask; index ncap = saturated)
int launch_sdf_green(hipStream_t s, uint8_t *masks, int n, int H, int W, uint8_t *gm, uint8_t *gb, int *cnt, const uint8_t *tab_out,
                     const uint8_t *tab_in, int ncap);
