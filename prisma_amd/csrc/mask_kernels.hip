// HBM-bound kernels of the mask_mmdet band (SOLOv2) for gfx950.
// Reference being replaced (paths under bands/mmdet/ of the reference):
//   datasets/pipelines/transforms.py:215-240,679-712,580-655  Resize(keep_ratio) / Normalize / Pad  -> mask_prep
//   models/backbones/resnet.py:612-627                        stem max-pool                          -> maxpool3x3s2
//   models/necks/fpn.py:161-188                               nearest top-down add, stride-2 extra level
//   core/utils/misc.py:190-208                                generate_coordinate                    -> coord_concat
//   models/dense_heads/solov2_head.py:134-150,253-292         bilinear resizes, GroupNorm + ReLU
//   models/dense_heads/solov2_head.py:616-623                 sigmoid + points NMS                   -> cls_points_nms
//   models/dense_heads/solov2_head.py:718-735                 mask threshold / area / maskness       -> mask_stats
//   core/post_processing/matrix_nms.py:62-98                  Matrix NMS decay                       -> matrix_nms
//   solov2_head.py:748-759 + bands/mask_mmdet.py:43-61,139-147  upsample x4, crop, resize, threshold, accumulate
#include "mask_kernels.h"

#include <algorithm>

namespace {

inline unsigned nblk(int64_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }
#define LAUNCH_CHECK()            \
    PB_HIP(hipGetLastError()); \
    return 0

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// torch upsample_bilinear2d source index (align_corners = False): src = max(scale * (dst + 0.5) - 0.5, 0)
__device__ __forceinline__ void lerp_src(int dst, float scale, int in, int &i0, int &i1, float &l1) {
    const float src = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

// Split maps (PB_PREC_SPLIT, mask_engine.hip): every pixel holds [hi (C) | lo (C)] fp16 pairs, value = hi + lo, lo = the fp16 rounding
// residual of the fp32 value.  The kernels below take `lo` = the half offset from an element to its residual; 0 = a plain fp16 map.
__device__ __forceinline__ void ld8(const f16 *p, int lo, float v[8]) {
    const f16x8 h = *(const f16x8 *)p;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
    if (lo) {
        const f16x8 l = *(const f16x8 *)(p + lo);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)l[j];
    }
}
__device__ __forceinline__ void st8(f16 *p, int lo, const float v[8]) {
    f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = (f16)v[j]; l[j] = (f16)(v[j] - (float)h[j]); }
    *(f16x8 *)p = h;
    if (lo) *(f16x8 *)(p + lo) = l;
}

// ------------------------------------------------------------------------------------------------
// frame prep: uint8 RGB [n][H][W][3] -> OpenCV 8-bit INTER_LINEAR resize to (nh, nw) (11-bit fixed-point
// coefficients, tables from the host) -> (v - mean) * (1 / std) in fp32 -> zero pad to (Hp, Wp) ->
// fp16, 4 x 4 pixel blocks as 64 channels ((dy * 4 + dx) * 4 + c, channel 3 = 0) of a [n][Hp / 4][Wp / 4] map: the layout in which the
// 7x7 / stride-2 stem is a 3x3 convolution with 4 x 64 output channels (mask_engine.hip).  One thread per padded pixel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_prep_kernel(const uint8_t *__restrict__ frames, int n, int H, int W, int nh, int nw,
                                                        int Hp, int Wp, const int4 *__restrict__ xt,
                                                        const int4 *__restrict__ yt, f16 *__restrict__ out,
                                                        float *__restrict__ chw, int lo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * Hp * Wp) return;
    const int x = (int)(i % Wp), y = (int)((i / Wp) % Hp), f = (int)(i / ((int64_t)Wp * Hp));
    f16x4 o, ol;
    o[0] = o[1] = o[2] = o[3] = (f16)0.f;
    ol = o;
    float v[3] = {0.f, 0.f, 0.f};
    if (y < nh && x < nw) {
        const int4 tx = xt[x], ty = yt[y];
        const uint8_t *r0 = frames + ((int64_t)f * H + ty.x) * W * 3, *r1 = frames + ((int64_t)f * H + ty.y) * W * 3;
        const float mean[3] = {123.675f, 116.28f, 103.53f};
        const float istd[3] = {(float)(1.0 / 58.395), (float)(1.0 / 57.12), (float)(1.0 / 57.375)};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = r0[tx.x * 3 + c] * tx.z + r0[tx.y * 3 + c] * tx.w;
            const int h1 = r1[tx.x * 3 + c] * tx.z + r1[tx.y * 3 + c] * tx.w;
            int q = (((ty.z * (h0 >> 4)) >> 16) + ((ty.w * (h1 >> 4)) >> 16) + 2) >> 2;
            q = q < 0 ? 0 : (q > 255 ? 255 : q);
            v[c] = ex_fmul(ex_fsub((float)q, mean[c]), istd[c]);
            o[c] = (f16)v[c];
            ol[c] = (f16)(v[c] - (float)o[c]);
        }
    }
    f16 *dst = out + (((int64_t)f * (Hp >> 2) + (y >> 2)) * (Wp >> 2) + (x >> 2)) * (64 + lo) + ((y & 3) * 4 + (x & 3)) * 4;
    *(f16x4 *)dst = o;
    if (lo) *(f16x4 *)(dst + lo) = ol;
    if (chw)
#pragma unroll
        for (int c = 0; c < 3; ++c) chw[(((int64_t)f * 3 + c) * Hp + y) * Wp + x] = v[c];
}

// ------------------------------------------------------------------------------------------------
// max_pool2d(3, stride 2, padding 1), NHWC fp16, 8 channels per thread
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const f16 *__restrict__ x, f16 *__restrict__ y, int n, int H, int W,
                                                           int OH, int OW, int C8, int lo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * OH * OW * C8) return;
    const int c = (int)(i % C8);
    const int64_t pix = i / C8;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
    const int ld = C8 * 8 + lo;
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            if ((unsigned)ix >= (unsigned)W) continue;
            float v[8];
            ld8(x + (((int64_t)b * H + iy) * W + ix) * ld + c * 8, lo, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], v[j]);
        }
    }
    st8(y + pix * ld + c * 8, lo, m);
}

// dst[n][h][w][C] += src[n][sh][sw][C] at the torch 'nearest' source pixel min(floor(d * sh / h), sh - 1)
__global__ __launch_bounds__(256) void nearest_add_kernel(f16 *__restrict__ dst, const f16 *__restrict__ src, int n, int h, int w,
                                                          int sh, int sw, int C8, float fy, float fx, int lo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * h * w * C8) return;
    const int c = (int)(i % C8);
    const int64_t pix = i / C8;
    const int x = (int)(pix % w), y = (int)((pix / w) % h), b = (int)(pix / ((int64_t)w * h));
    const int sy = min((int)floorf((float)y * fy), sh - 1), sx = min((int)floorf((float)x * fx), sw - 1);
    const int ld = C8 * 8 + lo;
    float a[8], s[8];
    ld8(dst + pix * ld + c * 8, lo, a);
    ld8(src + (((int64_t)b * sh + sy) * sw + sx) * ld + c * 8, lo, s);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += s[j];
    st8(dst + pix * ld + c * 8, lo, a);
}

// max_pool2d(kernel 1, stride 2): out[y][x] = in[2y][2x]
__global__ __launch_bounds__(256) void subsample2_kernel(const f16 *__restrict__ x, f16 *__restrict__ y, int n, int H, int W, int OH,
                                                         int OW, int C8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * OH * OW * C8) return;
    const int c = (int)(i % C8);
    const int64_t pix = i / C8;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
    *(f16x8 *)(y + i * 8) = *(const f16x8 *)(x + (((int64_t)b * H + 2 * oy) * W + 2 * ox) * C8 * 8 + c * 8);
}

// torch.linspace(-1, 1, steps)[i] in float32
__device__ __forceinline__ float linspace_pm1(int i, int steps) {
    if (steps <= 1) return -1.f;
    const float step = 2.f / (float)(steps - 1);
    return i < steps / 2 ? -1.f + step * (float)i : 1.f - step * (float)(steps - 1 - i);
}

// [n][h][w][C] (ld ldi) -> [n][h][w][C + 64]: channels C, C+1 = (x, y) coordinates in [-1, 1], rest 0
// split maps: the input's residuals at +lo_in, the output is [hi (C + 64) | lo (C + 64)]
__global__ __launch_bounds__(256) void coord_concat_kernel(const f16 *__restrict__ x, f16 *__restrict__ y, int n, int h, int w,
                                                           int C8, int ldi, int lo_in) {
    const int O8 = C8 + 8;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * h * w * O8) return;
    const int c = (int)(i % O8);
    const int64_t pix = i / O8;
    float o[8];
    if (c < C8) {
        ld8(x + pix * ldi + c * 8, lo_in, o);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
        if (c == C8) {
            const int px = (int)(pix % w), py = (int)((pix / w) % h);
            o[0] = linspace_pm1(px, w);
            o[1] = linspace_pm1(py, h);
        }
    }
    const int lo_out = lo_in ? O8 * 8 : 0;
    st8(y + pix * (int64_t)(O8 * 8 + lo_out) + c * 8, lo_out, o);
}

// NHWC fp16 bilinear resize, align_corners = False, separate pixel strides, optional accumulate into y
__global__ __launch_bounds__(256) void bilinear_kernel(const f16 *__restrict__ x, f16 *__restrict__ y, int n, int H, int W, int OH,
                                                       int OW, int C8, int ldi, int ldo, float sy, float sx, int accumulate, int lo_in,
                                                       int lo_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * OH * OW * C8) return;
    const int c = (int)(i % C8);
    const int64_t pix = i / C8;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
    int y0, y1, x0, x1;
    float ly, lx;
    lerp_src(oy, sy, H, y0, y1, ly);
    lerp_src(ox, sx, W, x0, x1, lx);
    const f16 *base = x + (int64_t)b * H * W * ldi + c * 8;
    float v00[8], v01[8], v10[8], v11[8], o[8];
    ld8(base + ((int64_t)y0 * W + x0) * ldi, lo_in, v00);
    ld8(base + ((int64_t)y0 * W + x1) * ldi, lo_in, v01);
    ld8(base + ((int64_t)y1 * W + x0) * ldi, lo_in, v10);
    ld8(base + ((int64_t)y1 * W + x1) * ldi, lo_in, v11);
    const float hy = 1.f - ly, hx = 1.f - lx;
    f16 *dst = y + pix * ldo + c * 8;
    if (accumulate) ld8(dst, lo_out, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = hy * (hx * v00[j] + lx * v01[j]) + ly * (hx * v10[j] + lx * v11[j]);
        o[j] = accumulate ? o[j] + v : v;
    }
    st8(dst, lo_out, o);
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(32 groups, eps 1e-5) + ReLU: per-(sample, chunk of 256 pixels, channel) partial sums -> one wave per
// (sample, group) adds them in a fixed order -> per-(sample, channel) affine -> one fused apply pass.  No atomics, and the
// chunking depends on the map size only, so a frame's statistics (hence its mask image) do not depend on which other frames
// share the launch or on block scheduling.  x: [n][HW][ldc] fp16; part: [n][nchunk][C][2].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const f16 *__restrict__ x, int HW, int C8, int ldc, float *__restrict__ part,
                                                       int chunk, int lo) {
    __shared__ float red[256 * 16];
    const int b = blockIdx.y;
    const int c8 = threadIdx.x % C8, pl = threadIdx.x / C8, npl = blockDim.x / C8;
    const int p0 = blockIdx.x * chunk, p1 = min(p0 + chunk, HW);
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (pl < npl) {
        for (int p = p0 + pl; p < p1; p += npl) {
            float v[8];
            ld8(x + ((int64_t)b * HW + p) * ldc + c8 * 8, lo, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = v[j]; s[j] += f; q[j] += f * f; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x * 16 + j] = s[j]; red[threadIdx.x * 16 + 8 + j] = q[j]; }
    __syncthreads();
    if (pl == 0) {
        for (int o = 1; o < npl; ++o)
#pragma unroll
            for (int j = 0; j < 16; ++j) red[c8 * 16 + j] += red[(o * C8 + c8) * 16 + j];
        float *dst = part + (((int64_t)b * gridDim.x + blockIdx.x) * C8 * 8 + c8 * 8) * 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dst[j * 2 + 0] = red[c8 * 16 + j];
            dst[j * 2 + 1] = red[c8 * 16 + 8 + j];
        }
    }
}

// part [n][nchunk][C][2] -> affine [n][C][2] = (rstd_g * gamma_c, beta_c - mean_g * rstd_g * gamma_c); one wave per (sample, group):
// lane l adds the (chunk, channel) items l, l + 64, ... in that order, then a fixed butterfly
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float *__restrict__ part, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ aff, int n, int C,
                                                          int cpg, int nchunk, float inv_cnt) {
    const int lane = threadIdx.x & 63;
    const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);          // (sample, group)
    const int groups = C / cpg;
    if (wg >= n * groups) return;
    const int b = wg / groups, g0 = (wg - b * groups) * cpg;
    float s = 0.f, q = 0.f;
    for (int it = lane; it < nchunk * cpg; it += 64) {
        const int ch = it / cpg, k = it - ch * cpg;
        const float *p = part + (((int64_t)b * nchunk + ch) * C + g0 + k) * 2;
        s += p[0]; q += p[1];
    }
    s = wave_sum(s); q = wave_sum(q);
    const float mean = s * inv_cnt;
    const float var = fmaxf(q * inv_cnt - mean * mean, 0.f);
    const float rs = rsqrtf(var + 1e-5f);
    for (int k = lane; k < cpg; k += 64) {
        const int c = g0 + k;
        const float r = rs * gamma[c];
        aff[((int64_t)b * C + c) * 2] = r;
        aff[((int64_t)b * C + c) * 2 + 1] = beta[c] - mean * r;
    }
}

__global__ __launch_bounds__(256) void gn_apply_relu_kernel(const f16 *__restrict__ x, const float *__restrict__ aff,
                                                            f16 *__restrict__ y, int n, int HW, int C8, int ldc, int ldo, int lo_in,
                                                            int lo_out, int dup) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * HW * C8) return;
    const int c8 = (int)(i % C8);
    const int64_t pix = i / C8;
    const int b = (int)(pix / HW);
    float v[8];
    ld8(x + pix * ldc + c8 * 8, lo_in, v);
    const float *a = aff + ((int64_t)b * C8 * 8 + c8 * 8) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j] * a[j * 2] + a[j * 2 + 1], 0.f);
    f16 *dst = y + pix * ldo + c8 * 8;
    st8(dst, lo_out, v);
    if (dup) *(f16x8 *)(dst + dup) = *(const f16x8 *)dst;       // [hi | hi | lo] rows: the map is the B operand of a split GEMM (dynamic convolution)
}

// ------------------------------------------------------------------------------------------------
// class scores: sigmoid, then keep a cell only if it equals the max of the 2x2 window whose lower-right corner
// it is (max_pool2d(2, stride 1, padding 1)[:-1, :-1] == s).  logit: this level's [n][g*g][C]; score: all levels
// concatenated per frame [n][pts_total][C], this level's cells start at row `off`.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cls_points_nms_kernel(const float *__restrict__ logit, float *__restrict__ score, int n,
                                                             int pts_total, int off, int g, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * g * g * C) return;
    const int c = (int)(i % C);
    const int cell = (int)((i / C) % (g * g)), b = (int)(i / ((int64_t)C * g * g));
    const int y = cell / g, x = cell - y * g;
    const float *base = logit + (int64_t)b * g * g * C + c;                    // this level's logits: [n][g*g][C]
    const float s = sigmoidf_(base[(int64_t)cell * C]);
    float m = s;
    if (x > 0) m = fmaxf(m, sigmoidf_(base[(int64_t)(cell - 1) * C]));
    if (y > 0) m = fmaxf(m, sigmoidf_(base[(int64_t)(cell - g) * C]));
    if (x > 0 && y > 0) m = fmaxf(m, sigmoidf_(base[(int64_t)(cell - g - 1) * C]));
    score[((int64_t)b * pts_total + off + cell) * C + c] = m == s ? s : 0.f;
}

// dst[k][cols] (fp16) = src[idx[k]][cols] (fp32); rows k >= count are zero; split: rows are [hi (cols) | lo (cols)]
__global__ __launch_bounds__(256) void gather_rows_f16_kernel(const float *__restrict__ src, const int *__restrict__ idx,
                                                              f16 *__restrict__ dst, int count, int rows_pad, int cols, int split) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)rows_pad * cols) return;
    const int c = (int)(i % cols), k = (int)(i / cols);
    const float v = k < count ? src[(int64_t)idx[k] * cols + c] : 0.f;
    const f16 h = (f16)v;
    if (!split) { dst[i] = h; return; }
    dst[(int64_t)k * 2 * cols + c] = h;
    dst[(int64_t)k * 2 * cols + cols + c] = (f16)(v - (float)h);
}

// per candidate row of dynamic-conv logits [k][ld]: area = #(sigmoid > thr), soft = sum of sigmoid over those
__global__ __launch_bounds__(256) void mask_stats_kernel(const float *__restrict__ logit, int HW, int64_t ld, float thr,
                                                         float *__restrict__ out) {
    const float *row = logit + (int64_t)blockIdx.x * ld;
    float cnt = 0.f, soft = 0.f;
    for (int p = threadIdx.x * 4; p < HW; p += 1024) {
        const f32x4 v = *(const f32x4 *)(row + p);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = sigmoidf_(v[j]);
            if (s > thr) { cnt += 1.f; soft += s; }
        }
    }
    cnt = wave_sum(cnt);
    soft = wave_sum(soft);
    __shared__ float red[8];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = cnt; red[4 + (threadIdx.x >> 6)] = soft; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = red[0] + red[1] + red[2] + red[3];
        out[blockIdx.x * 2 + 1] = red[4] + red[5] + red[6] + red[7];
    }
}

// bits[k][HW / 64]: bit p of row k = sigmoid(logit[idx[k]][p]) > thr (one ballot per wave = one 64-bit word)
__global__ __launch_bounds__(256) void bitpack_rows_kernel(const float *__restrict__ logit, int64_t ld, const int *__restrict__ idx,
                                                           int HW, float thr, unsigned long long *__restrict__ bits) {
    const int k = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool on = p < HW && sigmoidf_(logit[(int64_t)idx[k] * ld + p]) > thr;
    const unsigned long long m = __ballot(on);
    if ((threadIdx.x & 63) == 0 && p < HW) bits[(int64_t)k * (HW / 64) + (p >> 6)] = m;
}

// inter[i][j] = |mask_i & mask_j| for i <= j (matrix_nms.py:66-67 computes it as a float matmul of the 0/1 masks;
// the popcount is the same integer).  16 x 16 pairs per block, 64-word chunks of both row groups staged in LDS.
__global__ __launch_bounds__(256) void mask_intersections_kernel(const unsigned long long *__restrict__ bits, int n, int words,
                                                                 float *__restrict__ inter, int ld) {
    if (blockIdx.y > blockIdx.x) return;                       // only the upper triangle is ever read
    __shared__ unsigned long long a[16][65], b[16][65];
    const int tj = threadIdx.x & 15, ti = threadIdx.x >> 4;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    int acc = 0;
    for (int w0 = 0; w0 < words; w0 += 64) {
        for (int t = threadIdx.x; t < 16 * 64; t += 256) {
            const int r = t >> 6, w = t & 63;
            const bool okw = w0 + w < words;
            a[r][w] = (okw && i0 + r < n) ? bits[(int64_t)(i0 + r) * words + w0 + w] : 0ull;
            b[r][w] = (okw && j0 + r < n) ? bits[(int64_t)(j0 + r) * words + w0 + w] : 0ull;
        }
        __syncthreads();
#pragma unroll 8
        for (int w = 0; w < 64; ++w) acc += __popcll(a[ti][w] & b[tj][w]);
        __syncthreads();
    }
    if (i0 + ti < n && j0 + tj < n) inter[(int64_t)(i0 + ti) * ld + j0 + tj] = (float)acc;
}

// Matrix NMS, gaussian kernel (matrix_nms.py:62-98) on the score-sorted masks.  One wave per column j, lanes over i.
//   pass 0: comp[j] = max_{i < j, label_i == label_j} iou_ij                 (compensate_iou)
//   pass 1: out[j] = score[j] * min_i exp(-sigma d_ij^2) / exp(-sigma comp_i^2), d_ij = iou_ij for i < j with equal
//           labels and 0 otherwise                                            (decay_coefficient)
__global__ __launch_bounds__(64) void matrix_nms_kernel(const float *__restrict__ inter, int ld, const float *__restrict__ area,
                                                        const int *__restrict__ label, const float *__restrict__ score, int n,
                                                        float sigma, float *__restrict__ comp, float *__restrict__ out, int pass) {
    const int j = blockIdx.x;
    const int lj = label[j];
    const float aj = area[j];
    if (pass == 0) {
        float c = 0.f;
        for (int i = threadIdx.x; i < j; i += 64)
            if (label[i] == lj) {
                const float in = inter[(int64_t)i * ld + j];
                c = fmaxf(c, in / (area[i] + aj - in));
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c = fmaxf(c, __shfl_xor(c, o));
        if (threadIdx.x == 0) comp[j] = c;
        return;
    }
    float coeff = INFINITY;
    for (int i = threadIdx.x; i < n; i += 64) {
        float d = 0.f;
        if (i < j && label[i] == lj) {
            const float in = inter[(int64_t)i * ld + j];
            d = in / (area[i] + aj - in);
        }
        const float c = comp[i];
        coeff = fminf(coeff, expf(-1.f * sigma * (d * d)) / expf(-1.f * sigma * (c * c)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) coeff = fminf(coeff, __shfl_xor(coeff, o));
    if (threadIdx.x == 0) out[j] = score[j] * coeff;
}

// sig[k][HW] = sigmoid(logit[idx[k]][:HW])
__global__ __launch_bounds__(256) void sigmoid_rows_kernel(const float *__restrict__ logit, int64_t ld, const int *__restrict__ idx,
                                                           int HW, float *__restrict__ sig) {
    const int k = blockIdx.y;
    const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= HW) return;
    const f32x4 v = *(const f32x4 *)(logit + (int64_t)idx[k] * ld + p);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = sigmoidf_(v[j]);
    *(f32x4 *)(sig + (int64_t)k * HW + p) = o;
}

// Final masks: F.interpolate(sig, x4)[:h, :w] -> F.interpolate(size = (H, W)) -> > thr, both bilinear with
// align_corners = False, evaluated per output pixel (4 x 4 taps).  use[k] != 0 marks the instances the band
// accumulates (kept class, score > 0.5 and > --confidence): out pixel = (255 * count) mod 256 in all three
// channels.  inst (optional): the per-instance boolean masks [k][H][W].
__global__ __launch_bounds__(256) void band_accumulate_kernel(const float *__restrict__ sig, int k, int fh, int fw, int h, int w,
                                                              int H, int W, float sy2, float sx2, float thr,
                                                              const uint8_t *__restrict__ use, uint8_t *__restrict__ out,
                                                              uint8_t *__restrict__ inst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int X = (int)(i % W), Y = (int)(i / W);
    int ya, yb, xa, xb;
    float ly2, lx2;
    lerp_src(Y, sy2, h, ya, yb, ly2);
    lerp_src(X, sx2, w, xa, xb, lx2);
    int ry[2][2], rx[2][2];
    float fy[2], fx[2];
    lerp_src(ya, 0.25f, fh, ry[0][0], ry[0][1], fy[0]);
    lerp_src(yb, 0.25f, fh, ry[1][0], ry[1][1], fy[1]);
    lerp_src(xa, 0.25f, fw, rx[0][0], rx[0][1], fx[0]);
    lerp_src(xb, 0.25f, fw, rx[1][0], rx[1][1], fx[1]);
    int count = 0;
    for (int m = 0; m < k; ++m) {
        if (!inst && !use[m]) continue;
        const float *s = sig + (int64_t)m * fh * fw;
        float up[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float hy = 1.f - fy[a], hx = 1.f - fx[b];
                up[a][b] = hy * (hx * s[ry[a][0] * fw + rx[b][0]] + fx[b] * s[ry[a][0] * fw + rx[b][1]]) +
                           fy[a] * (hx * s[ry[a][1] * fw + rx[b][0]] + fx[b] * s[ry[a][1] * fw + rx[b][1]]);
            }
        const float v = (1.f - ly2) * ((1.f - lx2) * up[0][0] + lx2 * up[0][1]) + ly2 * ((1.f - lx2) * up[1][0] + lx2 * up[1][1]);
        const bool on = v > thr;
        if (inst) inst[(int64_t)m * H * W + i] = on ? 1 : 0;
        if (on && use[m]) ++count;
    }
    if (out) {
        const uint8_t b = (uint8_t)((255 * count) & 255);
        out[i * 3] = b; out[i * 3 + 1] = b; out[i * 3 + 2] = b;
    }
}

}  // namespace

int launch_mask_prep(hipStream_t s, const uint8_t *frames, int n, int H, int W, int nh, int nw, int Hp, int Wp, const int *xt,
                     const int *yt, f16 *out, float *chw, int split) {
    hipLaunchKernelGGL(mask_prep_kernel, dim3(nblk((int64_t)n * Hp * Wp)), dim3(256), 0, s, frames, n, H, W, nh, nw, Hp, Wp,
                       (const int4 *)xt, (const int4 *)yt, out, chw, split ? 64 : 0);
    LAUNCH_CHECK();
}
int launch_maxpool3x3s2(hipStream_t s, const f16 *x, f16 *y, int n, int H, int W, int C, int split) {
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(nblk((int64_t)n * OH * OW * (C / 8))), dim3(256), 0, s, x, y, n, H, W, OH, OW, C / 8,
                       split ? C : 0);
    LAUNCH_CHECK();
}
int launch_nearest_add(hipStream_t s, f16 *dst, const f16 *src, int n, int h, int w, int sh, int sw, int C, int split) {
    hipLaunchKernelGGL(nearest_add_kernel, dim3(nblk((int64_t)n * h * w * (C / 8))), dim3(256), 0, s, dst, src, n, h, w, sh, sw, C / 8,
                       (float)sh / (float)h, (float)sw / (float)w, split ? C : 0);
    LAUNCH_CHECK();
}
int launch_subsample2(hipStream_t s, const f16 *x, f16 *y, int n, int H, int W, int C) {
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(subsample2_kernel, dim3(nblk((int64_t)n * OH * OW * (C / 8))), dim3(256), 0, s, x, y, n, H, W, OH, OW, C / 8);
    LAUNCH_CHECK();
}
int launch_coord_concat(hipStream_t s, const f16 *x, f16 *y, int n, int h, int w, int C, int ldi, int lo_in) {
    hipLaunchKernelGGL(coord_concat_kernel, dim3(nblk((int64_t)n * h * w * (C / 8 + 8))), dim3(256), 0, s, x, y, n, h, w, C / 8, ldi, lo_in);
    LAUNCH_CHECK();
}
int launch_bilinear(hipStream_t s, const f16 *x, f16 *y, int n, int H, int W, int OH, int OW, int C, int ldi, int ldo,
                    int accumulate, int lo_in, int lo_out) {
    PB_CHECK(C % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0, -1, "bilinear: C=%d ldi=%d ldo=%d must be multiples of 8", C, ldi, ldo);
    hipLaunchKernelGGL(bilinear_kernel, dim3(nblk((int64_t)n * OH * OW * (C / 8))), dim3(256), 0, s, x, y, n, H, W, OH, OW, C / 8, ldi,
                       ldo, (float)H / (float)OH, (float)W / (float)OW, accumulate, lo_in, lo_out);
    LAUNCH_CHECK();
}
int gn_chunks(int HW) { return (HW + GN_CHUNK - 1) / GN_CHUNK; }
int launch_gn_relu(hipStream_t s, const f16 *x, f16 *y, int n, int HW, int C, int ldc, int ldo, int groups, const float *gamma,
                   const float *beta, float *stats, float *aff, int lo_in, int lo_out, int dup) {
    const int C8 = C / 8;
    PB_CHECK(C % 8 == 0 && C8 <= 256 && 256 % C8 == 0 && C % groups == 0, -1, "group norm: C=%d groups=%d unsupported", C, groups);
    const int chunk = GN_CHUNK, nchunk = gn_chunks(HW);           // fixed: the partition must not depend on the batch
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, n), dim3(256), 0, s, x, HW, C8, ldc, stats, chunk, lo_in);
    const int cpg = C / groups;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((n * groups + 3) / 4), dim3(256), 0, s, stats, gamma, beta, aff, n, C, cpg, nchunk,
                       1.f / ((float)HW * (float)cpg));
    hipLaunchKernelGGL(gn_apply_relu_kernel, dim3(nblk((int64_t)n * HW * C8)), dim3(256), 0, s, x, aff, y, n, HW, C8, ldc, ldo, lo_in,
                       lo_out, dup);
    LAUNCH_CHECK();
}
int launch_cls_points_nms(hipStream_t s, const float *logit, float *score, int n, int pts_total, int off, int g, int C) {
    hipLaunchKernelGGL(cls_points_nms_kernel, dim3(nblk((int64_t)n * g * g * C)), dim3(256), 0, s, logit, score, n, pts_total, off, g, C);
    LAUNCH_CHECK();
}
int launch_gather_rows_f16(hipStream_t s, const float *src, const int *idx, f16 *dst, int count, int rows_pad, int cols, int split) {
    hipLaunchKernelGGL(gather_rows_f16_kernel, dim3(nblk((int64_t)rows_pad * cols)), dim3(256), 0, s, src, idx, dst, count, rows_pad, cols,
                       split);
    LAUNCH_CHECK();
}
int launch_mask_stats(hipStream_t s, const float *logit, int rows, int HW, int64_t ld, float thr, float *out) {
    PB_CHECK(HW % 4 == 0 && ld % 4 == 0, -1, "mask_stats: HW=%d must be a multiple of 4", HW);
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(mask_stats_kernel, dim3(rows), dim3(256), 0, s, logit, HW, ld, thr, out);
    LAUNCH_CHECK();
}
int launch_bitpack_rows(hipStream_t s, const float *logit, int64_t ld, const int *idx, int count, int HW, float thr,
                        unsigned long long *bits) {
    PB_CHECK(HW % 64 == 0, -1, "bitpack: HW=%d must be a multiple of 64", HW);
    if (count <= 0) return 0;
    hipLaunchKernelGGL(bitpack_rows_kernel, dim3(nblk(HW), count), dim3(256), 0, s, logit, ld, idx, HW, thr, bits);
    LAUNCH_CHECK();
}
int launch_mask_intersections(hipStream_t s, const unsigned long long *bits, int n, int words, float *inter, int ld) {
    const unsigned t = (unsigned)((n + 15) / 16);
    hipLaunchKernelGGL(mask_intersections_kernel, dim3(t, t), dim3(256), 0, s, bits, n, words, inter, ld);
    LAUNCH_CHECK();
}
int launch_matrix_nms(hipStream_t s, const float *inter, int ld, const float *area, const int *label, const float *score, int n,
                      float sigma, float *comp, float *out) {
    PB_CHECK(n > 0, -1, "matrix_nms: n=%d", n);
    hipLaunchKernelGGL(matrix_nms_kernel, dim3(n), dim3(64), 0, s, inter, ld, area, label, score, n, sigma, comp, out, 0);
    hipLaunchKernelGGL(matrix_nms_kernel, dim3(n), dim3(64), 0, s, inter, ld, area, label, score, n, sigma, comp, out, 1);
    LAUNCH_CHECK();
}
int launch_sigmoid_rows(hipStream_t s, const float *logit, int64_t ld, const int *idx, int count, int HW, float *sig) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(sigmoid_rows_kernel, dim3(nblk(HW / 4), count), dim3(256), 0, s, logit, ld, idx, HW, sig);
    LAUNCH_CHECK();
}
// ---- signed distance field of the band's id image -> its green channel (bands/mask_mmdet.py:64-69,150-152 of the reference) ----------
// getSDF: inside = luminance != 0; sdf = distance to the mask (outside it) - distance to the background (inside it), remapped by
// ((sdf + 127) / 255 - 0.25) * 2, clipped to [0, 1], inverted, x 255, truncated to uint8.  The remap saturates for sdf >= 64.25 and
// sdf <= -63.25, so only SQUARED distances n <= 4128 (an integer: pixel offsets) matter - and the byte is a function of (side, n)
// alone.  The host tabulates that function with the reference's own float64 expression (prisma_amd/engine.py sdf_tables); the two
// kernels below compute the exact integer n of the Euclidean distance transform inside a +-64 window: a column scan for the vertical
// distance to the nearest pixel of either class, then per pixel the minimum of dx^2 + g(x + dx)^2 over the row.  Bytes equal the host
// restatement's on every frame.  Degenerate frames (no pixel of the other class anywhere) follow the reference's snowy.generate_sdf: its
// unsigned transform starts from INF = 1e20, so the distance is ~1e10 and the remap saturates - G = 0 on a frame without a mask, 255 on
// an all-mask frame (scipy's feature transform would instead answer as if a pixel sat at (row -1, column 0)).
enum { SDF_R = 64, SDF_FAR = 255 };

__global__ __launch_bounds__(256) void sdf_columns_kernel(const uint8_t *__restrict__ masks, uint8_t *__restrict__ gm, uint8_t *__restrict__ gb,
                                                          int *__restrict__ cnt, int n, int H, int W) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)n * W) return;
    const int f = (int)(t / W), x = (int)(t - (int64_t)f * W);
    const uint8_t *m = masks + ((int64_t)f * H * W + x) * 3;
    uint8_t *pm = gm + (int64_t)f * H * W + x, *pb = gb + (int64_t)f * H * W + x;
    int dm = SDF_FAR, db = SDF_FAR, inside_px = 0;         // vertical distance to the last mask / background pixel seen
    for (int y = 0; y < H; ++y) {
        const uint8_t *q = m + (int64_t)y * W * 3;
        const bool in = (q[0] | q[1] | q[2]) != 0;
        inside_px += in;
        dm = in ? 0 : (dm < SDF_FAR ? dm + 1 : SDF_FAR);
        db = in ? (db < SDF_FAR ? db + 1 : SDF_FAR) : 0;
        pm[(int64_t)y * W] = (uint8_t)dm;
        pb[(int64_t)y * W] = (uint8_t)db;
    }
    dm = db = SDF_FAR;
    for (int y = H - 1; y >= 0; --y) {
        const int um = pm[(int64_t)y * W], ub = pb[(int64_t)y * W];
        dm = um == 0 ? 0 : (dm < SDF_FAR ? dm + 1 : SDF_FAR);
        db = ub == 0 ? 0 : (db < SDF_FAR ? db + 1 : SDF_FAR);
        if (dm < um) pm[(int64_t)y * W] = (uint8_t)dm;
        if (db < ub) pb[(int64_t)y * W] = (uint8_t)db;
    }
    if (inside_px) atomicAdd(cnt + f, inside_px);
}

__global__ __launch_bounds__(256) void sdf_rows_kernel(uint8_t *__restrict__ masks, const uint8_t *__restrict__ gm, const uint8_t *__restrict__ gb,
                                                       const int *__restrict__ cnt, const uint8_t *__restrict__ tab_out,
                                                       const uint8_t *__restrict__ tab_in, int ncap, int H, int W) {
    __shared__ uint8_t sm[256 + 2 * SDF_R], sb[256 + 2 * SDF_R];
    const int f = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * 256;
    const int64_t row = ((int64_t)f * H + y) * W;
    for (int i = threadIdx.x; i < 256 + 2 * SDF_R; i += 256) {
        const int x = x0 - SDF_R + i;
        const bool ok = x >= 0 && x < W;
        sm[i] = ok ? gm[row + x] : (uint8_t)SDF_FAR;
        sb[i] = ok ? gb[row + x] : (uint8_t)SDF_FAR;
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    if (x >= W) return;
    const bool in = sm[threadIdx.x + SDF_R] == 0;                  // a mask pixel: its own vertical distance to the mask is 0
    const uint8_t *g = in ? sb : sm;                               // distances to the OTHER class
    const int inside_px = cnt[f];
    int best = ncap;
    if (!(in ? inside_px == H * W : inside_px == 0)) {                // other class empty: saturated (snowy's INF), best stays ncap
#pragma unroll 4
        for (int dx = -SDF_R; dx <= SDF_R; ++dx) {
            const int gv = g[threadIdx.x + SDF_R + dx];
            const int v = dx * dx + gv * gv;                       // SDF_FAR^2 > ncap: far columns never win
            best = v < best ? v : best;
        }
    }
    masks[(row + x) * 3 + 1] = in ? tab_in[best] : tab_out[best];
}

int launch_sdf_green(hipStream_t s, uint8_t *masks, int n, int H, int W, uint8_t *gm, uint8_t *gb, int *cnt, const uint8_t *tab_out,
                     const uint8_t *tab_in, int ncap) {
    PB_CHECK(ncap > 0 && ncap < SDF_FAR * SDF_FAR && ncap <= (SDF_R + 1) * (SDF_R + 1), -1, "sdf: table of %d entries does not fit the +-%d window", ncap + 1, SDF_R);
    PB_HIP(hipMemsetAsync(cnt, 0, (size_t)n * sizeof(int), s));
    hipLaunchKernelGGL(sdf_columns_kernel, dim3(nblk((int64_t)n * W)), dim3(256), 0, s, masks, gm, gb, cnt, n, H, W);
    hipLaunchKernelGGL(sdf_rows_kernel, dim3((W + 255) / 256, H, n), dim3(256), 0, s, masks, gm, gb, cnt, tab_out, tab_in, ncap, H, W);
    LAUNCH_CHECK();
}

int launch_band_accumulate(hipStream_t s, const float *sig, int k, int fh, int fw, int h, int w, int H, int W, float thr,
                           const uint8_t *use, uint8_t *out, uint8_t *inst) {
    hipLaunchKernelGGL(band_accumulate_kernel, dim3(nblk((int64_t)H * W)), dim3(256), 0, s, sig, k, fh, fw, h, w, H, W,
                       (float)h / (float)H, (float)w / (float)W, thr, use, out, inst);
    LAUNCH_CHECK();
}
