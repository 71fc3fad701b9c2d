// Launchers of the ZoeDepth metric head's non-GEMM kernels (zoe_kernels.hip).
#pragma once
#include "common.h"

int launch_softplus(hipStream_t s, float *x, int64_t rows, int cols, int ld);
// netd[row] = relu(<act[row][0:32], w2> + b2): output_conv2.2 + ReLU on the stored 32-channel activation
int launch_dot32_relu(hipStream_t s, const f16 *act, int ld, const float *w2, float b2, float *out, int64_t rows);
// out = a + bilinear(src -> (H, W), align_corners=True), NHWC fp16, C % 8 == 0
int launch_bilerp_add(hipStream_t s, const f16 *a, const f16 *src, f16 *out, int n, int h, int w, int H, int W, int C, int lda,
                      int lds, int ldo);
// AttractorLayerUnnormed: bnew = b + mean_a inv_attractor(A_a - b), b = bilinear(bprev -> (H, W), align_corners=True)
int launch_attractor(hipStream_t s, const float *A, int ldA, int nA, const float *bprev, int h, int w, float *bnew, int n, int H,
                     int W, float alpha);
// rows of the conditional log-binomial MLP input: [act32 | rel | bilinear(emb, 128 ch) | zeros] -> fp16 [rows][192]
int launch_zoe_cat(hipStream_t s, const f16 *act, int ld_act, const float *rel, const f16 *emb, int ld_emb, int h, int w, f16 *out,
                   int n, int H, int W);
// softplus(pt) -> p, t -> log-binomial softmax over 64 bins x bilinear(bins) -> metric depth
int launch_logbinom_depth(hipStream_t s, const float *pt, int ld_pt, const float *bins, int h, int w, float *depth, int n, int H,
                          int W, float min_temp, float max_temp);
// Pillow resize of float32 maps: horizontal pass then vertical pass, double accumulation, float32 intermediate
int launch_pil_resize(hipStream_t s, const float *in, float *tmp, float *out, int n, int h, int w, int H, int W, const int *xb,
                      const double *xk, int xks, const int *yb, const double *yk, int yks);
