// Kernels of the ZoeDepth metric head (`depth_anything --metric`), all HBM bound.  Reference being replaced:
// bands/patchfusion/zoedepth/models/layers/attractor.py:44-56,168-208 (inv_attractor, AttractorLayerUnnormed),
// layers/dist_layers.py:29-117 (log_binom, LogBinomial, ConditionalLogBinomial), layers/localbins_layers.py:86-113,
// models/zoedepth/zoedepth_v1.py:160-205 (wiring), bands/depth_anything.py:117-119 (Pillow resize of the prediction).
#include "zoe_kernels.h"

namespace {
inline unsigned nblk(int64_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }
#define LAUNCH_CHECK()         \
    PB_HIP(hipGetLastError()); \
    return 0

// torch softplus (beta 1, threshold 20)
__device__ __forceinline__ float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// torch upsample_bilinear2d, align_corners = True: src = dst * (in - 1) / (out - 1)
__device__ __forceinline__ void ac_src(int dst, float scale, int in, int &i0, int &i1, float &l1) {
    const float src = scale * (float)dst;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
}
inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

__global__ __launch_bounds__(256) void softplus_kernel(float *__restrict__ x, int64_t rows, int cols, int ld) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    float *p = x + (i / cols) * ld + (i % cols);
    *p = softplusf_(*p);
}

__global__ __launch_bounds__(256) void dot32_relu_kernel(const f16 *__restrict__ act, int ld, const float *__restrict__ w2, float b2,
                                                         float *__restrict__ out, int64_t rows) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float s = b2;
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
        const f16x8 v = *(const f16x8 *)(act + r * ld + c8 * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)v[j] * w2[c8 * 8 + j];
    }
    out[r] = fmaxf(s, 0.f);
}

__global__ __launch_bounds__(256) void bilerp_add_kernel(const f16 *__restrict__ a, const f16 *__restrict__ src, f16 *__restrict__ out,
                                                         int n, int h, int w, int H, int W, int C8, int lda, int lds, int ldo,
                                                         float sy, float sx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * H * W * C8) return;
    const int c = (int)(i % C8);
    const int64_t pix = i / C8;
    const int X = (int)(pix % W), Y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    int y0, y1, x0, x1;
    float ly, lx;
    ac_src(Y, sy, h, y0, y1, ly);
    ac_src(X, sx, w, x0, x1, lx);
    const f16 *base = src + (int64_t)b * h * w * lds + c * 8;
    const f16x8 v00 = *(const f16x8 *)(base + ((int64_t)y0 * w + x0) * lds), v01 = *(const f16x8 *)(base + ((int64_t)y0 * w + x1) * lds);
    const f16x8 v10 = *(const f16x8 *)(base + ((int64_t)y1 * w + x0) * lds), v11 = *(const f16x8 *)(base + ((int64_t)y1 * w + x1) * lds);
    const f16x8 va = *(const f16x8 *)(a + pix * lda + c * 8);
    const float hy = 1.f - ly, hx = 1.f - lx;
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        o[j] = (f16)((float)va[j] + (hy * (hx * (float)v00[j] + lx * (float)v01[j]) + ly * (hx * (float)v10[j] + lx * (float)v11[j])));
    *(f16x8 *)(out + pix * ldo + c * 8) = o;
}

// one thread per (pixel, bin); bprev / bnew are fp32 [n][h][w][64]
__global__ __launch_bounds__(256) void attractor_kernel(const float *__restrict__ A, int ldA, int nA, const float *__restrict__ bprev, int h,
                                                        int w, float *__restrict__ bnew, int n, int H, int W, float sy, float sx,
                                                        float alpha) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * H * W * 64) return;
    const int k = (int)(i & 63);
    const int64_t pix = i >> 6;
    const int X = (int)(pix % W), Y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    int y0, y1, x0, x1;
    float ly, lx;
    ac_src(Y, sy, h, y0, y1, ly);
    ac_src(X, sx, w, x0, x1, lx);
    const float *base = bprev + (int64_t)b * h * w * 64 + k;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float bc = hy * (hx * base[((int64_t)y0 * w + x0) * 64] + lx * base[((int64_t)y0 * w + x1) * 64]) +
                     ly * (hx * base[((int64_t)y1 * w + x0) * 64] + lx * base[((int64_t)y1 * w + x1) * 64]);
    const float *a = A + pix * ldA;
    float s = 0.f;
    for (int t = 0; t < nA; ++t) {
        const float dx = a[t] - bc;
        s += dx / (1.f + alpha * (dx * dx));
    }
    bnew[i] = bc + s / (float)nA;
}

// one thread per (pixel, 8-channel group of the 192-wide row)
__global__ __launch_bounds__(256) void zoe_cat_kernel(const f16 *__restrict__ act, int ld_act, const float *__restrict__ rel,
                                                      const f16 *__restrict__ emb, int ld_emb, int h, int w, f16 *__restrict__ out, int n,
                                                      int H, int W, float sy, float sx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * H * W * 24) return;
    const int g = (int)(i % 24);
    const int64_t pix = i / 24;
    f16x8 o;
    if (g < 4) {
        o = *(const f16x8 *)(act + pix * ld_act + g * 8);
    } else {
        const int X = (int)(pix % W), Y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
        int y0, y1, x0, x1;
        float ly, lx;
        ac_src(Y, sy, h, y0, y1, ly);
        ac_src(X, sx, w, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const f16 *base = emb + (int64_t)b * h * w * ld_emb;
        // columns 33 .. 160 hold the 128 embedding channels: group g covers columns g*8 .. g*8+7 = channels g*8-33 ..
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = g * 8 + j, c = col - 33;
            float v = 0.f;
            if (col == 32) v = rel[pix];
            else if (c >= 0 && c < 128)
                v = hy * (hx * (float)base[((int64_t)y0 * w + x0) * ld_emb + c] + lx * (float)base[((int64_t)y0 * w + x1) * ld_emb + c]) +
                    ly * (hx * (float)base[((int64_t)y1 * w + x0) * ld_emb + c] + lx * (float)base[((int64_t)y1 * w + x1) * ld_emb + c]);
            o[j] = (f16)v;
        }
    }
    *(f16x8 *)(out + pix * 192 + g * 8) = o;
}

// one wave per pixel, lane = bin (64 bins)
__global__ __launch_bounds__(256) void logbinom_depth_kernel(const float *__restrict__ pt, int ld_pt, const float *__restrict__ bins, int h,
                                                             int w, float *__restrict__ depth, int n, int H, int W, float sy, float sx,
                                                             float min_temp, float max_temp) {
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= (int64_t)n * H * W) return;
    const int k = threadIdx.x & 63;
    const int X = (int)(pix % W), Y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    int y0, y1, x0, x1;
    float ly, lx;
    ac_src(Y, sy, h, y0, y1, ly);
    ac_src(X, sx, w, x0, x1, lx);
    const float *base = bins + (int64_t)b * h * w * 64 + k;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float center = hy * (hx * base[((int64_t)y0 * w + x0) * 64] + lx * base[((int64_t)y0 * w + x1) * 64]) +
                         ly * (hx * base[((int64_t)y1 * w + x0) * 64] + lx * base[((int64_t)y1 * w + x1) * 64]);
    const float *q = pt + pix * ld_pt;
    const float p0 = softplusf_(q[0]) + 1e-4f, p1 = softplusf_(q[1]) + 1e-4f;
    const float t0 = softplusf_(q[2]) + 1e-4f, t1 = softplusf_(q[3]) + 1e-4f;
    const float p = p0 / (p0 + p1);
    const float t = (max_temp - min_temp) * (t0 / (t0 + t1)) + min_temp;
    const float eps = 1e-4f;
    const float om = fminf(fmaxf(1.f - p, eps), 1.f), xp = fminf(fmaxf(p, eps), 1.f);
    // log_binom(K - 1, k) by Stirling, eps 1e-7 (dist_layers.py:29-33)
    const float nn = 63.f + 1e-7f, kk = (float)k + 1e-7f;
    const float lb = nn * logf(nn) - kk * logf(kk) - (nn - kk) * logf(nn - kk + 1e-7f);
    const float y = (lb + (float)k * logf(xp) + (float)(63 - k) * logf(om)) / t;
    float m = y;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float e = expf(y - m);
    float se = e, sc = e * center;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { se += __shfl_xor(se, o); sc += __shfl_xor(sc, o); }
    if (k == 0) depth[pix] = sc / se;
}

// Pillow ImagingResampleHorizontal_32bpc / Vertical_32bpc: bounds b[2*i] = first source index, b[2*i+1] = tap count
__global__ __launch_bounds__(256) void pil_h_kernel(const float *__restrict__ in, float *__restrict__ out, int n, int h, int w, int W,
                                                    const int *__restrict__ xb, const double *__restrict__ xk, int ks) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * h * W) return;
    const int X = (int)(i % W);
    const int64_t row = i / W;
    const float *src = in + row * w + xb[2 * X];
    const double *k = xk + (int64_t)X * ks;
    double ss = 0.0;
    for (int t = 0; t < xb[2 * X + 1]; ++t) ss = ex_dadd(ss, ex_dmul((double)src[t], k[t]));
    out[i] = (float)ss;
}
__global__ __launch_bounds__(256) void pil_v_kernel(const float *__restrict__ in, float *__restrict__ out, int n, int h, int H, int W,
                                                    const int *__restrict__ yb, const double *__restrict__ yk, int ks) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)n * H * W) return;
    const int X = (int)(i % W), Y = (int)((i / W) % H), b = (int)(i / ((int64_t)W * H));
    const float *src = in + ((int64_t)b * h + yb[2 * Y]) * W + X;
    const double *k = yk + (int64_t)Y * ks;
    double ss = 0.0;
    for (int t = 0; t < yb[2 * Y + 1]; ++t) ss = ex_dadd(ss, ex_dmul((double)src[(int64_t)t * W], k[t]));
    out[i] = (float)ss;
}
}  // namespace

int launch_softplus(hipStream_t s, float *x, int64_t rows, int cols, int ld) {
    hipLaunchKernelGGL(softplus_kernel, dim3(nblk(rows * cols)), dim3(256), 0, s, x, rows, cols, ld);
    LAUNCH_CHECK();
}
int launch_dot32_relu(hipStream_t s, const f16 *act, int ld, const float *w2, float b2, float *out, int64_t rows) {
    hipLaunchKernelGGL(dot32_relu_kernel, dim3(nblk(rows)), dim3(256), 0, s, act, ld, w2, b2, out, rows);
    LAUNCH_CHECK();
}
int launch_bilerp_add(hipStream_t s, const f16 *a, const f16 *src, f16 *out, int n, int h, int w, int H, int W, int C, int lda,
                      int lds, int ldo) {
    hipLaunchKernelGGL(bilerp_add_kernel, dim3(nblk((int64_t)n * H * W * (C / 8))), dim3(256), 0, s, a, src, out, n, h, w, H, W, C / 8, lda,
                       lds, ldo, ac_scale(h, H), ac_scale(w, W));
    LAUNCH_CHECK();
}
int launch_attractor(hipStream_t s, const float *A, int ldA, int nA, const float *bprev, int h, int w, float *bnew, int n, int H,
                     int W, float alpha) {
    hipLaunchKernelGGL(attractor_kernel, dim3(nblk((int64_t)n * H * W * 64)), dim3(256), 0, s, A, ldA, nA, bprev, h, w, bnew, n, H, W,
                       ac_scale(h, H), ac_scale(w, W), alpha);
    LAUNCH_CHECK();
}
int launch_zoe_cat(hipStream_t s, const f16 *act, int ld_act, const float *rel, const f16 *emb, int ld_emb, int h, int w, f16 *out,
                   int n, int H, int W) {
    hipLaunchKernelGGL(zoe_cat_kernel, dim3(nblk((int64_t)n * H * W * 24)), dim3(256), 0, s, act, ld_act, rel, emb, ld_emb, h, w, out, n,
                       H, W, ac_scale(h, H), ac_scale(w, W));
    LAUNCH_CHECK();
}
int launch_logbinom_depth(hipStream_t s, const float *pt, int ld_pt, const float *bins, int h, int w, float *depth, int n, int H,
                          int W, float min_temp, float max_temp) {
    hipLaunchKernelGGL(logbinom_depth_kernel, dim3(nblk((int64_t)n * H * W, 4)), dim3(256), 0, s, pt, ld_pt, bins, h, w, depth, n, H, W,
                       ac_scale(h, H), ac_scale(w, W), min_temp, max_temp);
    LAUNCH_CHECK();
}
int launch_pil_resize(hipStream_t s, const float *in, float *tmp, float *out, int n, int h, int w, int H, int W, const int *xb,
                      const double *xk, int xks, const int *yb, const double *yk, int yks) {
    // Pillow skips a pass whose size does not change
    const float *cur = in;
    if (W != w) {
        float *dst = H != h ? tmp : out;
        hipLaunchKernelGGL(pil_h_kernel, dim3(nblk((int64_t)n * h * W)), dim3(256), 0, s, cur, dst, n, h, w, W, xb, xk, xks);
        cur = dst;
    }
    if (H != h) hipLaunchKernelGGL(pil_v_kernel, dim3(nblk((int64_t)n * H * W)), dim3(256), 0, s, cur, out, n, h, H, W, yb, yk, yks);
    else if (W == w) PB_HIP(hipMemcpyAsync(out, in, (size_t)n * h * w * 4, hipMemcpyDeviceToDevice, s));
    LAUNCH_CHECK();
}
