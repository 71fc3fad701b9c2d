// Launchers of the flow_raft band's non-GEMM kernels (raft_kernels.hip).
#pragma once
#include "common.h"

int launch_raft_prep(hipStream_t s, const uint8_t *frames, int F, int H, int W, int sh, int sw, int Hp, int Wp, int pad_l,
                     int pad_t, int resize, const int *xi, const int *xc, const int *yi, const int *yc, f16 *out,
                     uint8_t *scaled_out, int s2d = 0, int lo_off = 0, int lo8_pa = -1, int norm_mode = 0,      // norm_mode 1: ImageNet mean / std (GMFlow)
                     int isz = 0);     // isz: (Hp, Wp) is flow_gmflow's --inference_size - bilinear (align_corners) of the scaled frame, no padding
// ld: row stride of `out` in halfs (0 = Kp); o8: also store the row as e4m3 (unscaled) after its Kp halfs (gemm.h nk16)
int launch_im2col7_flow(hipStream_t s, const float *x, int B, int H, int W, f16 *out, int Kp, int ld = 0, int o8 = 0);
int in_stats_chunks(int HW);
int launch_in_stats(hipStream_t s, const f16 *x, int B, int HW, int C, int ldc, float *part, float *stats, int lo_off = 0, int lo8_pa = -1);
int launch_in_apply(hipStream_t s, const f16 *a, const float *sa, const f16 *b, const float *sb, f16 *out, int B, int HW,
                    int C, int ldc, int lo_off = 0, int lo8_pa = -1);
// ld: pixel stride of hx / hx2 (384, or 576 with the fp8 copy at byte o8_off = 768, scaled by o8_scale)
int launch_init_state(hipStream_t s, const f16 *c, float *h32, f16 *hx, f16 *hx2, float *flow, int64_t rows, int ld = 384, int o8_off = 0,
                      float o8_scale = 16.f, int inp_off = 128);      // inp_off: channel offset of the context features inside hx / hx2
int launch_avgpool2_nhwc(hipStream_t s, const f16 *x, f16 *y, int n, int H, int W, int C);
int launch_corr_tile(hipStream_t s, const f16 *x, f16 *y, int F, int h, int w, int wp, int npad);
// volume.hip: out[M, N] fp16 = A[M, 256] . W[N, 256]^T (one pair, one pyramid level); W has w_rows >= N addressable rows; the
// A-stationary persistent kernel for this K = 256, output-bound shape
int launch_corr_volume(hipStream_t s, const f16 *A, int M, const f16 *W, int N, int w_rows, f16 *out, int64_t ldo);
int launch_corr_lookup(hipStream_t s, const f16 *const lv[4], const int h[4], const int w[4], const int wp[4], const int ld[4],
                       const float *flow, int P, int w8, f16 *out, int64_t rows, int ldo = 384, int o8_off = 0, float o8_scale = 16.f);
// convf1 (7 x 7, 2 -> 128, ReLU) straight from the fp32 flow field (raft_kernels.hip convf1_kernel)
int convf1_packed_halfs(int passes);
void convf1_pack(const float *w, int passes, f16 *dst);
int launch_convf1(hipStream_t s, const float *flow, const f16 *wpk, const float *bias, f16 *out, int64_t rows, int P, int h8, int w8, int ldo,
                  int o8_off, float o8_scale, int passes);
int launch_flow_head2(hipStream_t s, const f16 *x, const f16 *w, const float *bias, float *flow, int n, int H, int W, int split = 0);
int launch_put_flow(hipStream_t s, const float *flow, f16 *hx, f16 *hx2, int64_t rows, int ld = 384, int o8_off = 0, float o8_scale = 16.f,
                    int flow_off = 382);       // flow_off: channel offset of the two flow channels inside hx / hx2
// flow_gmflow --inference_size: flow [N, ih, iw, 2] -> [N, sh, sw, 2], bilinear (align_corners), u * sw / iw, v * sh / ih; maxd as launch_upsample
int launch_flow_resize_back(hipStream_t s, const float *in, int N, int ih, int iw, int sh, int sw, float *out, unsigned *maxd);
int launch_upsample(hipStream_t s, const float *flow, const float *mask, int N, int h8, int w8, int pad_l, int pad_t, int sh,
                    int sw, float *out, unsigned *maxd);
int launch_flow_encode(hipStream_t s, const float *flow, int N, int sh, int sw, const unsigned *maxd, uint8_t *rgb,
                       float *max_out);
int launch_fwdbwd_mask(hipStream_t s, const float *flow, int n, int h, int w, float a1, float a2, uint8_t *mask);
