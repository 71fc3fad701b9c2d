// HBM-bound kernels of the depth_anything band for gfx950: pre-process, LayerNorm, bilinear
// resizes, per-frame min/max, heat-map encode, layout converters.  All of them are one pass over
// their data with 8/16-byte per-lane accesses; none is reshaped into a GEMM.
#include "kernels.h"

namespace {

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per token row (D <= 1024, D % 4 == 0), fp32 in, fp16 out.
// reference: nn.LayerNorm(eps=1e-6) in dinov2 blocks (block.py:82-107) and the final norm
// applied to the 4 taps (vision_transformer.py:297-321).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                        const float *__restrict__ bt, f16 *__restrict__ y, int B,
                                                        int ntp, int ntok, int D, float eps, int drop_cls, int ldy,
                                                        int lo_off, int o8_off, float o8_scale, int lo8_pa) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= (int64_t)B * ntok) return;
    const int b = (int)(r / ntok), t = (int)(r - (int64_t)b * ntok);
    if (drop_cls && t == 0) return;
    const float *xr = x + ((int64_t)b * ntp + t) * D;
    f16 *yr = drop_cls ? y + ((int64_t)b * (ntok - 1) + (t - 1)) * ldy : y + ((int64_t)b * ntp + t) * ldy;
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < D) {
            v[j] = *(const f32x4 *)(xr + c);
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[j][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < D) {
            const f32x4 gg = *(const f32x4 *)(g + c), bb = *(const f32x4 *)(bt + c);
            f16x4 o, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = (v[j][e] - mean) * rstd * gg[e] + bb[e];
                o[e] = (f16)t;
                l[e] = (f16)(t - (float)o[e]);
            }
            *(f16x4 *)(yr + c) = o;
            if (lo_off && lo8_pa >= 0) {                       // split map with e4m3 residual parts: [hi | hi8 | lo8] (gemm.h lo8)
                const float shi = __builtin_ldexpf(1.f, lo8_pa), slo = __builtin_ldexpf(1.f, lo8_pa + 12);
                float tv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) tv[e] = (v[j][e] - mean) * rstd * gg[e] + bb[e];
                *(int *)((char *)yr + 2 * lo_off + c) = pb_fp8x4((float)o[0] * shi, (float)o[1] * shi, (float)o[2] * shi, (float)o[3] * shi);
                *(int *)((char *)yr + 3 * lo_off + c) = pb_fp8x4((tv[0] - (float)o[0]) * slo, (tv[1] - (float)o[1]) * slo, (tv[2] - (float)o[2]) * slo,
                                                                 (tv[3] - (float)o[3]) * slo);
            } else if (lo_off) *(f16x4 *)(yr + c + lo_off) = l;       // split-fp16 consumers read [hi | lo] (gemm.h)
            if (o8_off)                                        // fp8 copy after the row's fp16 part (gemm.h nk16)
                *(int *)((char *)yr + o8_off + c) = pb_fp8x4((float)o[0] * o8_scale, (float)o[1] * o8_scale, (float)o[2] * o8_scale, (float)o[3] * o8_scale);
        }
    }
}

__global__ void cls_rows_kernel(float *resid, const float *cls, const float *pos, int B, int ntp, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, c = i - b * D;
    resid[(int64_t)b * ntp * D + c] = cls[c] + pos[c];
}

// ------------------------------------------------------------------------------------------------
// Pre-process (bands/depth_anything.py:122-126): /255 -> cv2.resize(INTER_CUBIC) -> normalise ->
// CHW, fused with the im2col of the 14x14/s14 patch embedding: thread per network pixel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t *__restrict__ frames, int B, int H, int W,
                                                         int nh, int nw, const int *__restrict__ xi,
                                                         const float *__restrict__ xw, const int *__restrict__ yi,
                                                         const float *__restrict__ yw, f16 *__restrict__ out, int Kp,
                                                         float *__restrict__ chw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * nh * nw) return;
    const int x = (int)(i % nw);
    const int y = (int)((i / nw) % nh);
    const int b = (int)(i / ((int64_t)nw * nh));
    const uint8_t *img = frames + (int64_t)b * H * W * 3;
    int xs[4];
    float wx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { xs[t] = xi[x * 4 + t] * 3; wx[t] = xw[x * 4 + t]; }
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) {
        const uint8_t *row = img + (int64_t)yi[y * 4 + ty] * W * 3;
        float r[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int tx = 0; tx < 4; ++tx)
#pragma unroll
            for (int c = 0; c < 3; ++c) r[c] += (float)row[xs[tx] + c] * wx[tx];
        const float wy = yw[y * 4 + ty];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += r[c] * wy;
    }
    const float mean[3] = {0.485f, 0.456f, 0.406f}, istd[3] = {1.f / 0.229f, 1.f / 0.224f, 1.f / 0.225f};
    const int gw = nw / 14;
    const int gy = y / 14, py = y - gy * 14, gx = x / 14, px = x - gx * 14;
    const int64_t prow = ((int64_t)b * (nh / 14) * gw + gy * gw + gx) * Kp;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = (acc[c] * (1.f / 255.f) - mean[c]) * istd[c];
        if (out) out[prow + c * 196 + py * 14 + px] = (f16)v;
        if (chw) chw[((int64_t)(b * 3 + c) * nh + y) * nw + x] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Bilinear resize, NHWC fp16, 8 channels (16 B) per thread.  torch upsample_bilinear2d semantics.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bilerp_src(int dst, float scale, int in, int align, int &i0, int &i1, float &l1) {
    float src = align ? scale * (float)dst : fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
}
__host__ __device__ inline float bilerp_scale(int in, int out, int align) {
    if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    return (float)in / (float)out;
}

__global__ __launch_bounds__(256) void bilinear_nhwc_kernel(const f16 *__restrict__ x, f16 *__restrict__ y, int B,
                                                            int H, int W, int OH, int OW, int C8, int ldc, int align,
                                                            float sy, float sx, int lo_off, int lo8_pa) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * OH * OW * C8) return;
    const int c = (int)(i % C8);
    const int64_t pix = i / C8;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
    int y0, y1, x0, x1;
    float ly, lx;
    bilerp_src(oy, sy, H, align, y0, y1, ly);
    bilerp_src(ox, sx, W, align, x0, x1, lx);
    const f16 *base = x + (int64_t)b * H * W * ldc + c * 8;
    const f16x8 v00 = *(const f16x8 *)(base + ((int64_t)y0 * W + x0) * ldc);
    const f16x8 v01 = *(const f16x8 *)(base + ((int64_t)y0 * W + x1) * ldc);
    const f16x8 v10 = *(const f16x8 *)(base + ((int64_t)y1 * W + x0) * ldc);
    const f16x8 v11 = *(const f16x8 *)(base + ((int64_t)y1 * W + x1) * ldc);
    const float hy = 1.f - ly, hx = 1.f - lx;
    f16x8 o;
    if (lo_off && lo8_pa >= 0) {     // split maps with e4m3 residual parts [hi | hi8 | lo8] (gemm.h lo8)
        const float shi = __builtin_ldexpf(1.f, lo8_pa), slo = __builtin_ldexpf(1.f, lo8_pa + 12), inv = __builtin_ldexpf(1.f, -(lo8_pa + 12));
        const int64_t o00 = ((int64_t)y0 * W + x0) * ldc, o01 = ((int64_t)y0 * W + x1) * ldc, o10 = ((int64_t)y1 * W + x0) * ldc,
                      o11 = ((int64_t)y1 * W + x1) * ldc;
        const char *bb = (const char *)(x + (int64_t)b * H * W * ldc) + 3 * lo_off + c * 8;
        const int64_t po[4] = {o00, o01, o10, o11};
        float lo[4][8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int2 u = *(const int2 *)(bb + po[t] * 2);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int wv = h ? u.y : u.x;
                const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(wv, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(wv, true);
                lo[t][4 * h + 0] = a[0] * inv; lo[t][4 * h + 1] = a[1] * inv; lo[t][4 * h + 2] = d[0] * inv; lo[t][4 * h + 3] = d[1] * inv;
            }
        }
        float tv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = (float)v00[j] + lo[0][j], b2 = (float)v01[j] + lo[1][j], c2 = (float)v10[j] + lo[2][j], d = (float)v11[j] + lo[3][j];
            tv[j] = hy * (hx * a + lx * b2) + ly * (hx * c2 + lx * d);
            o[j] = (f16)tv[j];
        }
        f16 *dst = y + pix * ldc;
        *(f16x8 *)(dst + c * 8) = o;
        int2 h8, l8;
        h8.x = pb_fp8x4((float)o[0] * shi, (float)o[1] * shi, (float)o[2] * shi, (float)o[3] * shi);
        h8.y = pb_fp8x4((float)o[4] * shi, (float)o[5] * shi, (float)o[6] * shi, (float)o[7] * shi);
        l8.x = pb_fp8x4((tv[0] - (float)o[0]) * slo, (tv[1] - (float)o[1]) * slo, (tv[2] - (float)o[2]) * slo, (tv[3] - (float)o[3]) * slo);
        l8.y = pb_fp8x4((tv[4] - (float)o[4]) * slo, (tv[5] - (float)o[5]) * slo, (tv[6] - (float)o[6]) * slo, (tv[7] - (float)o[7]) * slo);
        *(int2 *)((char *)dst + 2 * lo_off + c * 8) = h8;
        *(int2 *)((char *)dst + 3 * lo_off + c * 8) = l8;
        return;
    }
    if (lo_off) {        // split-fp16 maps [hi | lo]: interpolate hi + lo, store the result's hi and lo
        const f16x8 l00 = *(const f16x8 *)(base + ((int64_t)y0 * W + x0) * ldc + lo_off);
        const f16x8 l01 = *(const f16x8 *)(base + ((int64_t)y0 * W + x1) * ldc + lo_off);
        const f16x8 l10 = *(const f16x8 *)(base + ((int64_t)y1 * W + x0) * ldc + lo_off);
        const f16x8 l11 = *(const f16x8 *)(base + ((int64_t)y1 * W + x1) * ldc + lo_off);
        f16x8 l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = (float)v00[j] + (float)l00[j], b2 = (float)v01[j] + (float)l01[j];
            const float c2 = (float)v10[j] + (float)l10[j], d = (float)v11[j] + (float)l11[j];
            const float t = hy * (hx * a + lx * b2) + ly * (hx * c2 + lx * d);
            o[j] = (f16)t;
            l[j] = (f16)(t - (float)o[j]);
        }
        *(f16x8 *)(y + pix * ldc + c * 8) = o;
        *(f16x8 *)(y + pix * ldc + c * 8 + lo_off) = l;
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
        o[j] = (f16)(hy * (hx * (float)v00[j] + lx * (float)v01[j]) + ly * (hx * (float)v10[j] + lx * (float)v11[j]));
    *(f16x8 *)(y + pix * ldc + c * 8) = o;
}

// ------------------------------------------------------------------------------------------------
// Band tail (bands/depth_anything.py:132, 215-216): bilinear(align_corners=False) to the frame
// size + per-frame min/max via order-preserving uint atomics.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__device__ __forceinline__ void block_minmax(float lo, float hi, unsigned *mm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(mm, f2ord(lo));
        atomicMax(mm + 1, f2ord(hi));
    }
}

__global__ __launch_bounds__(256) void depth_resize_minmax_kernel(const float *__restrict__ net, int nh, int nw,
                                                                  float *__restrict__ out, int H, int W, float sy,
                                                                  float sx, unsigned *mm) {
    const int b = blockIdx.y;
    const float *src = net + (int64_t)b * nh * nw;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)H * W;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % W), oy = (int)(i / W);
        int y0, y1, x0, x1;
        float ly, lx;
        bilerp_src(oy, sy, nh, 0, y0, y1, ly);
        bilerp_src(ox, sx, nw, 0, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float v = hy * (hx * src[y0 * nw + x0] + lx * src[y0 * nw + x1]) +
                        ly * (hx * src[y1 * nw + x0] + lx * src[y1 * nw + x1]);
        out[(int64_t)b * H * W + i] = v;
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    block_minmax(lo, hi, mm + 2 * b);
}

__global__ __launch_bounds__(256) void minmax_kernel(const float *__restrict__ x, int64_t per, unsigned *mm) {
    const int b = blockIdx.y;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[(int64_t)b * per + i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    block_minmax(lo, hi, mm + 2 * b);
}

__global__ void init_minmax_kernel(unsigned *mm, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { mm[2 * i] = 0xFFFFFFFFu; mm[2 * i + 1] = 0u; }
}

// heat_to_rgb(1 - (p - min) / (max - min)) * 255 -> uint8 (truncation).
// bands/depth_anything.py:215-220 + bands/common/encode.py:13-33: normalise / flip in float32,
// colour ramp in float64 with numpy's operation order; explicit _rn intrinsics keep the compiler
// from contracting mul+add into FMAs so that the bytes equal the reference's.
__global__ __launch_bounds__(256) void heat_encode_kernel(const float *__restrict__ depth, int64_t per,
                                                          const unsigned *__restrict__ mm, int flip,
                                                          uint8_t *__restrict__ rgb, float *mn, float *mx) {
    const int b = blockIdx.y;
    const float dmin = ord2f(mm[2 * b]), dmax = ord2f(mm[2 * b + 1]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (mn) mn[b] = dmin;
        if (mx) mx[b] = dmax;
    }
    if (!rgb) return;
    const float range = ex_fsub(dmax, dmin);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
        float d = ex_fdiv(ex_fsub(depth[(int64_t)b * per + i], dmin), range);
        if (flip) d = ex_fsub(1.0f, d);
        const double hue = ex_dmul(ex_dsub(1.0, (double)d), 0.65);
        const double h6 = ex_dmul(hue, 6.0);
        const double off[3] = {0.0, 4.0, 2.0};
        uint8_t o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = ex_dadd(h6, off[c]);
            v = fmod(v, 6.0);
            if (v < 0.0) v = ex_dadd(v, 6.0);
            v = ex_dsub(fabs(ex_dsub(v, 3.0)), 1.0);
            v = fmin(fmax(v, 0.0), 1.0);
            v = ex_dmul(v, 255.0);
            o[c] = (v == v) ? (uint8_t)(int)v : (uint8_t)0;      // NaN (max == min) -> 0 like numpy on x86
        }
        uint8_t *dst = rgb + ((int64_t)b * per + i) * 3;
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
    }
}

// ---- still-image / --subpath encode: write_depth(heatmap=True) ---------------------------------------------------
// bands/common/io.py:138-172 + encode.py:73-95 (float_to_edge, saturation), :141-146 (float_to_rgb): normalise (float32), flip,
// q = uint8(d * 255) (truncation), Sobel ksize 1 = central differences of q with a reflect-101 border, edge = mag * (255 / max mag) / 255
// in float64, heat ramp in float64, rgb = rgb * (1 - edge) + edge, min / max packed as 24-bit fixed point of [0, 1000] into pixels
// (0, 0) and (0, 1), uint8 truncation.  Three passes after the min / max reduction: quantise, max |gradient|^2 (integers: exact, and
// sqrt is monotone, so the max commutes with it), encode.  Every floating operation is separately rounded (common.h ex_*).
__device__ __forceinline__ float still_norm(float v, float dmin, float range, int flip) {
    float d = ex_fdiv(ex_fsub(v, dmin), range);
    if (flip) d = ex_fsub(1.0f, d);
    return d;
}

__global__ __launch_bounds__(256) void still_quant_kernel(const float *__restrict__ depth, int64_t per, const unsigned *__restrict__ mm,
                                                          int flip, uint8_t *__restrict__ q) {
    const float dmin = ord2f(mm[0]), range = ex_fsub(ord2f(mm[1]), dmin);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = ex_fmul(still_norm(depth[i], dmin, range, flip), 255.0f);
        q[i] = (v == v) ? (uint8_t)(int)v : (uint8_t)0;
    }
}

__device__ __forceinline__ int still_grad2(const uint8_t *__restrict__ q, int H, int W, int y, int x) {
    // cv2.Sobel(img, CV_64F, 1, 0, ksize=1) / (0, 1): [-1, 0, 1], BORDER_REFLECT_101 (index -1 -> 1, n -> n - 2; a 1-pixel axis has no gradient)
    const int xm = x > 0 ? x - 1 : (W > 1 ? 1 : 0), xp = x + 1 < W ? x + 1 : (W > 1 ? W - 2 : 0);
    const int ym = y > 0 ? y - 1 : (H > 1 ? 1 : 0), yp = y + 1 < H ? y + 1 : (H > 1 ? H - 2 : 0);
    const int gx = (int)q[(int64_t)y * W + xp] - (int)q[(int64_t)y * W + xm];
    const int gy = (int)q[(int64_t)yp * W + x] - (int)q[(int64_t)ym * W + x];
    return gx * gx + gy * gy;
}

__global__ __launch_bounds__(256) void still_gradmax_kernel(const uint8_t *__restrict__ q, int H, int W, unsigned *g2max) {
    int best = 0;
    const int64_t per = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = still_grad2(q, H, W, (int)(i / W), (int)(i % W));
        best = g > best ? g : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(best, o); best = t > best ? t : best; }
    if ((threadIdx.x & 63) == 0 && best > 0) atomicMax(g2max, (unsigned)best);
}

// float_to_rgb(value, 0, 1000) * 255 -> uint8 with numpy >= 2 promotion (a float32 scalar stays float32 against Python scalars)
__device__ __forceinline__ void still_range_px(float value, uint8_t *dst) {
    float t = ex_fdiv(ex_fsub(value, 0.0f), ex_fsub(1000.0f, 0.0f));
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float L = ex_fmul(t, 16777215.0f);
    const float c[3] = {ex_fdiv(floorf(fmodf(L, 256.0f)), 255.0f), ex_fdiv(fmodf(floorf(ex_fdiv(L, 256.0f)), 256.0f), 255.0f),
                        ex_fdiv(fmodf(floorf(ex_fdiv(L, 65536.0f)), 256.0f), 255.0f)};
#pragma unroll
    for (int k = 0; k < 3; ++k) dst[k] = (uint8_t)(int)ex_dmul((double)c[k], 255.0);
}

__global__ __launch_bounds__(256) void still_encode_kernel(const float *__restrict__ depth, const uint8_t *__restrict__ q, int H, int W,
                                                           const unsigned *__restrict__ mm, const unsigned *__restrict__ g2max, int flip,
                                                           int encode_range, uint8_t *__restrict__ rgb, float *mnmx) {
    const float dmin = ord2f(mm[0]), dmax = ord2f(mm[1]), range = ex_fsub(dmax, dmin);
    if (blockIdx.x == 0 && threadIdx.x == 0 && mnmx) { mnmx[0] = dmin; mnmx[1] = dmax; }
    const double magmax = sqrt((double)*g2max);
    const double gain = ex_ddiv(255.0, magmax);                      // 255 / 0 = inf: a flat image encodes NaN -> 0 like the reference
    const int64_t per = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
        uint8_t *dst = rgb + i * 3;
        if (encode_range && i < 2) { still_range_px(i == 0 ? dmin : dmax, dst); continue; }
        const float d = still_norm(depth[i], dmin, range, flip);
        const double mag = sqrt((double)still_grad2(q, H, W, (int)(i / W), (int)(i % W)));
        const double edge = ex_ddiv(ex_dmul(mag, gain), 255.0);
        const double sat = ex_dsub(1.0, edge), inv = ex_dsub(1.0, sat);
        const double h6 = ex_dmul(ex_dmul(ex_dsub(1.0, (double)d), 0.65), 6.0);
        const double off[3] = {0.0, 4.0, 2.0};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = fmod(ex_dadd(h6, off[c]), 6.0);
            if (v < 0.0) v = ex_dadd(v, 6.0);
            v = ex_dsub(fabs(ex_dsub(v, 3.0)), 1.0);
            v = fmin(fmax(v, 0.0), 1.0);
            v = ex_dmul(ex_dadd(ex_dmul(v, sat), inv), 255.0);
            dst[c] = (v == v) ? (uint8_t)(int)v : (uint8_t)0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// layout converters (tests / stage dumps)
// ------------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float *x, f16 *y, int B, int C, int H, int W, int ldc, int relu) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * C * H * W) return;
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    const int64_t hw = (int64_t)H * W;
    const int b = (int)(pix / hw);
    const int64_t p = pix - b * hw;
    float v = x[((int64_t)b * C + c) * hw + p];
    if (relu) v = fmaxf(v, 0.f);
    y[pix * ldc + c] = (f16)v;
}
__global__ void nhwc_to_nchw_kernel(const f16 *x, float *y, int B, int C, int H, int W, int ldc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * C * H * W) return;
    const int64_t hw = (int64_t)H * W;
    const int64_t p = i % hw;
    const int c = (int)((i / hw) % C);
    const int b = (int)(i / (hw * C));
    y[i] = (float)x[((int64_t)b * hw + p) * ldc + c];
}
__global__ void f32_to_f16_kernel(const float *x, f16 *y, int64_t rows, int cols, int ld_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    y[r * ld_out + c] = (f16)x[i];
}
__global__ void f16_to_f32_kernel(const f16 *x, float *y, int64_t rows, int cols, int ld_in) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    y[i] = (float)x[r * ld_in + c];
}

__global__ void fill_random_f16_kernel(f16 *x, int64_t n, unsigned seed, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        x[i] = (f16)(((float)(h >> 8) * (1.f / 8388608.f) - 1.f) * scale);
    }
}

inline unsigned nblk(int64_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }

}  // namespace

int launch_layernorm(hipStream_t s, const float *x, const float *g, const float *b, f16 *y, int B, int ntp, int ntok,
                     int D, float eps, int drop_cls, int ldy, int lo_off, int o8_off, float o8_scale, int lo8_pa) {
    PB_CHECK(D % 4 == 0 && D <= 1024, -1, "layernorm: D=%d unsupported", D);
    hipLaunchKernelGGL(layernorm_kernel, dim3(nblk((int64_t)B * ntok, 4)), dim3(256), 0, s, x, g, b, y, B, ntp, ntok,
                       D, eps, drop_cls, ldy ? ldy : D, lo_off, o8_off, o8_scale, lo8_pa);
    PB_HIP(hipGetLastError());
    return 0;
}

int launch_cls_rows(hipStream_t s, float *resid, const float *cls, const float *pos, int B, int ntp, int D) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3(nblk((int64_t)B * D)), dim3(256), 0, s, resid, cls, pos, B, ntp, D);
    PB_HIP(hipGetLastError());
    return 0;
}

int launch_preprocess(hipStream_t s, const uint8_t *frames, int B, int H, int W, int nh, int nw, const int *xi,
                      const float *xw, const int *yi, const float *yw, f16 *out, int Kp, float *chw_out) {
    hipLaunchKernelGGL(preprocess_kernel, dim3(nblk((int64_t)B * nh * nw)), dim3(256), 0, s, frames, B, H, W, nh, nw,
                       xi, xw, yi, yw, out, Kp, chw_out);
    PB_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// DPT head tail at the network resolution, from a LOW-resolution map (dpt.py:158-160: output_conv1 -> F.interpolate(bilinear, align_corners=True) ->
// output_conv2 = conv3x3(F/2 -> 32), ReLU, conv1x1(32 -> 1), ReLU).  A convolution is linear and the resize is linear, so the 3 x 3 convolution's nine
// tap products W[t] y can be taken BEFORE the resize: z[t * 32 + c] = sum_k W[c, k, t] y[k] is one 1 x 1 GEMM over output_conv1's map (288 columns, at
// a third of the pixels), and a network pixel p is
//     depth(p) = relu(b2 + sum_c w2[c] relu(bias[c] + sum_t [p + t inside the upsampled map] bilerp(z[t * 32 + c]; p + t))),
// the zero padding of the convolution being "the tap's sample lies outside".  Against resizing to a [hi | hi8 | lo8] map of 128 channels (7.8 GB
// written and read back per 32 frames) and an implicit GEMM with N = 32 over it: a third of the matrix work, a third of the bytes, and no rounding of the
// upsampled map at all (the bilinear weights meet fp32 values).  4 lanes x 8 channels per pixel, 16 consecutive pixels of a row per wave, 4 rows per block.
// z: [B, H, W] pixels of ldz halfs - plain fp16 (lo_off = 0), [hi | lo] fp16 pairs (lo8_pa < 0) or [hi | hi8 | lo8] (gemm.h lo8).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dpt_tail_kernel(const f16 *__restrict__ z, int H, int W, int ldz, int lo_off, int lo8_pa, float sy, float sx,
                                                       const float *__restrict__ bias, const float *__restrict__ w2, float b2,
                                                       float *__restrict__ out, int OH, int OW) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q4 = lane & 3, px = lane >> 2;
    const int X = blockIdx.x * 16 + px, Y = blockIdx.y * 4 + wave, b = blockIdx.z;
    const bool live = X < OW && Y < OH;
    const int Xc = live ? X : 0, Yc = live ? Y : 0;
    // the three rows / columns the taps sample: source cells and weights of F.interpolate(align_corners=True)
    int y0[3], y1[3], x0[3], x1[3];
    float ly[3], lx[3];
    bool oky[3], okx[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int qy = Yc + t - 1, qx = Xc + t - 1;
        oky[t] = qy >= 0 && qy < OH; okx[t] = qx >= 0 && qx < OW;
        bilerp_src(oky[t] ? qy : 0, sy, H, 1, y0[t], y1[t], ly[t]);
        bilerp_src(okx[t] ? qx : 0, sx, W, 1, x0[t], x1[t], lx[t]);
    }
    // one buffer resource per image: the loads below carry a 32-bit lane offset (a flat load spends two 64-bit adds per address, and there are 72 of them)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(z + (int64_t)b * H * W * ldz, (unsigned)((int64_t)H * W * ldz * 2));
    const int lane_ch = q4 * 8;
    typedef int i32x4v __attribute__((ext_vector_type(4)));
    typedef int i32x2v __attribute__((ext_vector_type(2)));
    const float inv = __builtin_ldexpf(1.f, -(lo8_pa + 12));
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            if (!(oky[ky] && okx[kx])) continue;
            const int ch = (ky * 3 + kx) * 32;
            const int o[4] = {(y0[ky] * W + x0[kx]) * ldz, (y0[ky] * W + x1[kx]) * ldz, (y1[ky] * W + x0[kx]) * ldz, (y1[ky] * W + x1[kx]) * ldz};
            // the four cells' weights once per tap, then acc += w hi + (w 2^-(pa + 12)) lo8 per cell: two FMAs and half a conversion per (cell, channel)
            const float hy = 1.f - ly[ky], hx = 1.f - lx[kx];
            const float w4[4] = {hy * hx, hy * lx[kx], ly[ky] * hx, ly[ky] * lx[kx]};
            f16x8 h[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) h[n] = __builtin_bit_cast(f16x8, (i32x4v)__builtin_amdgcn_raw_buffer_load_b128(rs, (o[n] + ch + lane_ch) * 2, 0, 0));
            if (lo_off && lo8_pa >= 0) {
                int2 u[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const i32x2v t2 = (i32x2v)__builtin_amdgcn_raw_buffer_load_b64(rs, o[n] * 2 + 3 * lo_off + ch + lane_ch, 0, 0);
                    u[n].x = t2[0]; u[n].y = t2[1];
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const float wl = w4[n] * inv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += w4[n] * (float)h[n][j];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int wv = hh ? u[n].y : u[n].x;
                        const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(wv, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(wv, true);
                        acc[4 * hh + 0] += wl * a[0]; acc[4 * hh + 1] += wl * a[1]; acc[4 * hh + 2] += wl * d[0]; acc[4 * hh + 3] += wl * d[1];
                    }
                }
            } else if (lo_off) {
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const f16x8 l = __builtin_bit_cast(f16x8, (i32x4v)__builtin_amdgcn_raw_buffer_load_b128(rs, (o[n] + ch + lane_ch + lo_off) * 2, 0, 0));
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += w4[n] * ((float)h[n][j] + (float)l[j]);
                }
            } else {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += w4[n] * (float)h[n][j];
            }
        }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += fmaxf(acc[j] + bias[q4 * 8 + j], 0.f) * w2[q4 * 8 + j];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (live && q4 == 0) out[((int64_t)b * OH + Y) * OW + X] = fmaxf(s + b2, 0.f);
}

int launch_dpt_tail(hipStream_t s, const f16 *z, int B, int H, int W, int ldz, int lo_off, int lo8_pa, const float *bias, const float *w2, float b2,
                    float *out, int OH, int OW) {
    PB_CHECK(ldz % 8 == 0 && lo_off % 8 == 0 && (int64_t)H * W * ldz * 2 < (1LL << 31), -1, "dpt tail: pixel stride %d / residual offset %d must be multiples of 8, an image below 2 GB", ldz, lo_off);
    const float sy = bilerp_scale(H, OH, 1), sx = bilerp_scale(W, OW, 1);
    hipLaunchKernelGGL(dpt_tail_kernel, dim3((OW + 15) / 16, (OH + 3) / 4, B), dim3(256), 0, s, z, H, W, ldz, lo_off, lo8_pa, sy, sx, bias, w2, b2, out, OH, OW);
    PB_HIP(hipGetLastError());
    return 0;
}

int launch_bilinear_nhwc(hipStream_t s, const f16 *x, f16 *y, int B, int H, int W, int OH, int OW, int C, int ldc,
                         int align, int lo_off, int lo8_pa) {
    PB_CHECK(C % 8 == 0 && ldc % 8 == 0, -1, "bilinear: C=%d ldc=%d must be multiples of 8", C, ldc);
    const float sy = bilerp_scale(H, OH, align), sx = bilerp_scale(W, OW, align);
    hipLaunchKernelGGL(bilinear_nhwc_kernel, dim3(nblk((int64_t)B * OH * OW * (C / 8))), dim3(256), 0, s, x, y, B, H,
                       W, OH, OW, C / 8, ldc, align, sy, sx, lo_off, lo8_pa);
    PB_HIP(hipGetLastError());
    return 0;
}

int launch_init_minmax(hipStream_t s, unsigned *mm, int B) {
    hipLaunchKernelGGL(init_minmax_kernel, dim3(nblk(B)), dim3(256), 0, s, mm, B);
    PB_HIP(hipGetLastError());
    return 0;
}

int launch_depth_resize_minmax(hipStream_t s, const float *net, int B, int nh, int nw, float *out, int H, int W,
                               unsigned *mm) {
    const float sy = bilerp_scale(nh, H, 0), sx = bilerp_scale(nw, W, 0);
    unsigned gx = nblk((int64_t)H * W);
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(depth_resize_minmax_kernel, dim3(gx, B), dim3(256), 0, s, net, nh, nw, out, H, W, sy, sx, mm);
    PB_HIP(hipGetLastError());
    return 0;
}

int launch_minmax_only(hipStream_t s, const float *x, int B, int64_t per, unsigned *mm) {
    unsigned gx = nblk(per);
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(minmax_kernel, dim3(gx, B), dim3(256), 0, s, x, per, mm);
    PB_HIP(hipGetLastError());
    return 0;
}

int launch_heat_encode(hipStream_t s, const float *depth, int B, int H, int W, const unsigned *mm, int flip,
                       uint8_t *rgb, float *mn, float *mx) {
    unsigned gx = nblk((int64_t)H * W);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(heat_encode_kernel, dim3(gx, B), dim3(256), 0, s, depth, (int64_t)H * W, mm, flip, rgb, mn, mx);
    PB_HIP(hipGetLastError());
    return 0;
}

int launch_still_encode(hipStream_t s, const float *depth, int H, int W, unsigned *mm, unsigned *g2max, uint8_t *q, int flip, int encode_range,
                        uint8_t *rgb, float *mnmx) {
    const int64_t per = (int64_t)H * W;
    unsigned gx = nblk(per);
    if (gx > 1024) gx = 1024;
    PB_HIP(hipMemsetAsync(g2max, 0, 4, s));
    hipLaunchKernelGGL(init_minmax_kernel, dim3(1), dim3(64), 0, s, mm, 1);
    hipLaunchKernelGGL(minmax_kernel, dim3(gx > 512 ? 512 : gx, 1), dim3(256), 0, s, depth, per, mm);
    hipLaunchKernelGGL(still_quant_kernel, dim3(gx), dim3(256), 0, s, depth, per, mm, flip, q);
    hipLaunchKernelGGL(still_gradmax_kernel, dim3(gx), dim3(256), 0, s, q, H, W, g2max);
    hipLaunchKernelGGL(still_encode_kernel, dim3(gx), dim3(256), 0, s, depth, q, H, W, mm, g2max, flip, encode_range, rgb, mnmx);
    PB_HIP(hipGetLastError());
    return 0;
}

int launch_nchw_f32_to_nhwc_f16(hipStream_t s, const float *x, f16 *y, int B, int C, int H, int W, int ldc, int relu) {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(nblk((int64_t)B * C * H * W)), dim3(256), 0, s, x, y, B, C, H, W, ldc,
                       relu);
    PB_HIP(hipGetLastError());
    return 0;
}
int launch_nhwc_f16_to_nchw_f32(hipStream_t s, const f16 *x, float *y, int B, int C, int H, int W, int ldc) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblk((int64_t)B * C * H * W)), dim3(256), 0, s, x, y, B, C, H, W,
                       ldc);
    PB_HIP(hipGetLastError());
    return 0;
}
int launch_f32_to_f16(hipStream_t s, const float *x, f16 *y, int64_t rows, int cols, int ld_out) {
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3(nblk(rows * cols)), dim3(256), 0, s, x, y, rows, cols, ld_out);
    PB_HIP(hipGetLastError());
    return 0;
}
int launch_f16_to_f32(hipStream_t s, const f16 *x, float *y, int64_t rows, int cols, int ld_in) {
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3(nblk(rows * cols)), dim3(256), 0, s, x, y, rows, cols, ld_in);
    PB_HIP(hipGetLastError());
    return 0;
}

int launch_fill_random_f16(hipStream_t s, f16 *x, int64_t n, unsigned seed, float scale) {
    hipLaunchKernelGGL(fill_random_f16_kernel, dim3(4096), dim3(256), 0, s, x, n, seed, scale);
    PB_HIP(hipGetLastError());
    return 0;
}
