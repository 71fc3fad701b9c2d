// HBM / gather bound kernels of the flow_raft band for gfx950.
// Reference being replaced: bands/flow_raft.py:99-107 (resize, pad), bands/raft/extractor.py (InstanceNorm),
// bands/raft/corr.py:12-50 (pyramid + 9x9 lookup), bands/raft/update.py:33-60 (GRU gating),
// bands/raft/raft.py:73-84 (convex upsample), bands/common/encode.py:98-126 (process_flow).
#include "raft_kernels.h"

#include <algorithm>

namespace {

inline unsigned nblk(int64_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }

// ------------------------------------------------------------------------------------------------
// frame prep: uint8 RGB frame -> [optional 8-bit fixed-point bicubic resize] -> replicate pad to /8 ->
// 2*(x/255)-1 -> fp16 [F][Hp][Wp][4] (channel 3 = 0).  One thread per padded pixel.
// ------------------------------------------------------------------------------------------------
// mode 0: RAFT 2 (x / 255) - 1 (raft.py:91-92); mode 1: GMFlow (x / 255 - mean) / std with the ImageNet statistics (gmflow/utils.py:53-58)
__device__ __forceinline__ float prep_norm(float v, int c, int mode) {
    if (mode == 0) return 2.f * (v / 255.f) - 1.f;
    const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f), sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
    return (v / 255.f - mean) / sd;
}

// pixel (sy, sx) of the SCALED frame: cv2.resize(fx = fy = scale, INTER_CUBIC) on uint8 (fixed-point taps, common/flow.py / flow_raft.py:100)
// or the frame itself
__device__ __forceinline__ void scaled_px(const uint8_t *img, int W, int sy, int sx, int resize, const int *__restrict__ xi,
                                          const int *__restrict__ xc, const int *__restrict__ yi, const int *__restrict__ yc, int (&v)[3]) {
    if (resize) {
        int acc[3] = {0, 0, 0};
#pragma unroll
        for (int ty = 0; ty < 4; ++ty) {
            const uint8_t *row = img + (int64_t)yi[sy * 4 + ty] * W * 3;
            int r[3] = {0, 0, 0};
#pragma unroll
            for (int tx = 0; tx < 4; ++tx) {
                const int o = xi[sx * 4 + tx] * 3, c = xc[sx * 4 + tx];
                r[0] += row[o] * c; r[1] += row[o + 1] * c; r[2] += row[o + 2] * c;
            }
            const int cy = yc[sy * 4 + ty];
            acc[0] += r[0] * cy; acc[1] += r[1] * cy; acc[2] += r[2] * cy;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int q = (acc[c] + (1 << 21)) >> 22;
            v[c] = q < 0 ? 0 : (q > 255 ? 255 : q);
        }
    } else {
        const uint8_t *px = img + ((int64_t)sy * W + sx) * 3;
        v[0] = px[0]; v[1] = px[1]; v[2] = px[2];
    }
}

__global__ __launch_bounds__(256) void raft_prep_kernel(const uint8_t *__restrict__ frames, int F, int H, int W, int sh,
                                                        int sw, int Hp, int Wp, int pad_l, int pad_t, int resize,
                                                        const int *__restrict__ xi, const int *__restrict__ xc,
                                                        const int *__restrict__ yi, const int *__restrict__ yc,
                                                        f16 *__restrict__ out, uint8_t *__restrict__ scaled_out, int s2d, int lo_off,
                                                        int lo8_pa, int norm_mode, int isz) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)F * Hp * Wp) return;
    const int x = (int)(i % Wp), y = (int)((i / Wp) % Hp), f = (int)(i / ((int64_t)Wp * Hp));
    const uint8_t *img = frames + (int64_t)f * H * W * 3;
    int v[3];
    float vf[3];
    if (isz) {
        // flow_gmflow --inference_size (reference flow_gmflow.py:76-80): the network input is F.interpolate(bilinear, align_corners = True) of
        // the scaled frame as float - no padding; torch's arithmetic: src = dst * (in - 1) / (out - 1), weights (1 - l, l), rows of the
        // horizontal blends
        const float ry = Hp > 1 ? (float)(sh - 1) / (float)(Hp - 1) : 0.f, rx = Wp > 1 ? (float)(sw - 1) / (float)(Wp - 1) : 0.f;
        const float fy = ry * (float)y, fx = rx * (float)x;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < sh - 1 ? 1 : 0), x1 = x0 + (x0 < sw - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        int p00[3], p01[3], p10[3], p11[3];
        scaled_px(img, W, y0, x0, resize, xi, xc, yi, yc, p00);
        scaled_px(img, W, y0, x1, resize, xi, xc, yi, yc, p01);
        scaled_px(img, W, y1, x0, resize, xi, xc, yi, yc, p10);
        scaled_px(img, W, y1, x1, resize, xi, xc, yi, yc, p11);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            vf[c] = (1.f - ly) * ((1.f - lx) * (float)p00[c] + lx * (float)p01[c]) + ly * ((1.f - lx) * (float)p10[c] + lx * (float)p11[c]);
            v[c] = 0;
        }
    } else {
        int sy = y - pad_t, sx = x - pad_l;                   // position in the scaled (unpadded) frame
        sy = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        sx = sx < 0 ? 0 : (sx > sw - 1 ? sw - 1 : sx);
        scaled_px(img, W, sy, sx, resize, xi, xc, yi, yc, v);
#pragma unroll
        for (int c = 0; c < 3; ++c) vf[c] = (float)v[c];
    }
    f16x4 o, l;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float t = prep_norm(vf[c], c, norm_mode);
        o[c] = (f16)t;
        l[c] = (f16)(t - (float)o[c]);
    }
    o[3] = (f16)0.f; l[3] = (f16)0.f;
    // s2d: 4 x 4 pixel blocks become the 64 channels ((dy * 4 + dx) * 4 + c) of a [F, Hp / 4, Wp / 4] map - the layout in which the
    // 7x7 / stride-2 stem is a 3x3 convolution with 4 x 64 output channels (raft_engine.hip)
    if (s2d && lo_off) {        // split-fp16 encoder input: [hi (64) | lo (64)] per space-to-depth pixel
        const int64_t px = (((int64_t)f * (Hp >> 2) + (y >> 2)) * (Wp >> 2) + (x >> 2)) * (2 * lo_off) + ((y & 3) * 4 + (x & 3)) * 4;
        *(f16x4 *)(out + px) = o;
        if (lo8_pa >= 0) {      // e4m3 residual parts: [hi | hi8 | lo8] per pixel (gemm.h lo8)
            const float shi = __builtin_ldexpf(1.f, lo8_pa), slo = __builtin_ldexpf(1.f, lo8_pa + 12);
            float t[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) t[c] = prep_norm(vf[c], c, norm_mode);
            char *pb = (char *)(out + px - ((y & 3) * 4 + (x & 3)) * 4);
            const int col = ((y & 3) * 4 + (x & 3)) * 4;
            *(int *)(pb + 2 * lo_off + col) = pb_fp8x4((float)o[0] * shi, (float)o[1] * shi, (float)o[2] * shi, 0.f);
            *(int *)(pb + 3 * lo_off + col) = pb_fp8x4((t[0] - (float)o[0]) * slo, (t[1] - (float)o[1]) * slo, (t[2] - (float)o[2]) * slo, 0.f);
        } else {
            *(f16x4 *)(out + px + lo_off) = l;
        }
    } else {
        const int64_t oi = s2d ? ((((int64_t)f * (Hp >> 2) + (y >> 2)) * (Wp >> 2) + (x >> 2)) * 16 + (y & 3) * 4 + (x & 3)) : i;
        *(f16x4 *)(out + oi * 4) = o;
    }
    if (scaled_out && !isz && y >= pad_t && y < pad_t + sh && x >= pad_l && x < pad_l + sw) {
        uint8_t *d = scaled_out + (((int64_t)f * sh + (y - pad_t)) * sw + (x - pad_l)) * 3;
        d[0] = (uint8_t)v[0]; d[1] = (uint8_t)v[1]; d[2] = (uint8_t)v[2];
    }
}

// ------------------------------------------------------------------------------------------------
// im2col for the 7x7 convolution of the 2-channel flow (BasicMotionEncoder.convf1; the image stems are space-to-depth convs):
// rows = output pixels, k = (ky*7 + kx)*C + c, zero padded to Kp.  One thread per (row, tap).
// ------------------------------------------------------------------------------------------------
template <typename T, int CS>
__global__ __launch_bounds__(256) void im2col7_kernel(const T *__restrict__ x, int B, int H, int W, int C, int stride,
                                                       int OH, int OW, f16 *__restrict__ out, int Kp, int ld, int o8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * OH * OW * 49) return;
    const int tap = (int)(i % 49);
    const int64_t row = i / 49;
    const int ox = (int)(row % OW), oy = (int)((row / OW) % OH), b = (int)(row / ((int64_t)OW * OH));
    const int ky = tap / 7, kx = tap - ky * 7;
    const int iy = oy * stride - 3 + ky, ix = ox * stride - 3 + kx;
    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const T *src = x + (((int64_t)b * H + iy) * W + ix) * CS;
    f16 *dst = out + row * ld + tap * C;
    for (int c = 0; c < C; ++c) dst[c] = ok ? (f16)(float)src[c] : (f16)0.f;
    if (o8) {                // fp8 copy (unscaled: flow is in pixels) after the row's Kp halfs (gemm.h nk16)
        unsigned char *d8 = (unsigned char *)(out + row * ld + Kp) + tap * C;
        for (int c = 0; c < C; c += 2) {
            const unsigned short u = pb_fp8x2(ok ? (float)dst[c] : 0.f, ok && c + 1 < C ? (float)dst[c + 1] : 0.f);
            d8[c] = (unsigned char)(u & 0xFF);
            if (c + 1 < C) d8[c + 1] = (unsigned char)(u >> 8);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// InstanceNorm2d (no affine, eps 1e-5, biased variance over H x W per sample and channel).
// instance norm (extractor.py norm_fn='instance': nn.InstanceNorm2d defaults - biased variance, eps 1e-5, no affine).
// Pass 1: every block sums one pixel chunk of one image and writes its {sum, sum of squares} per channel to part[chunk][b][c]
// - no atomics, so the result does not depend on block scheduling.  Pass 2: one thread per (b, c) adds the chunks in order and
// stores {mean, 1 / sqrt(var + eps)}.  Threads cover (pixel lane, 8 channels).
// ------------------------------------------------------------------------------------------------
// 8 e4m3 residuals of a split map's pixel ([hi | hi8 | lo8], gemm.h lo8) -> float
__device__ __forceinline__ void lo8_load8(const char *p, float inv, float (&l)[8]) {
    const int2 u = *(const int2 *)p;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int wv = h ? u.y : u.x;
        const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8(wv, false), d = __builtin_amdgcn_cvt_pk_f32_fp8(wv, true);
        l[4 * h + 0] = a[0] * inv; l[4 * h + 1] = a[1] * inv; l[4 * h + 2] = d[0] * inv; l[4 * h + 3] = d[1] * inv;
    }
}

// lo_off != 0: x is a split-fp16 map [hi | lo] and the statistics are those of hi + lo.
__global__ __launch_bounds__(256) void in_stats_kernel(const f16 *__restrict__ x, int HW, int C8, int ldc,
                                                        float *__restrict__ part, int chunk, int lo_off, int lo8_pa) {
    __shared__ float red[256 * 16];
    const int b = blockIdx.y;
    const int c8 = threadIdx.x % C8, pl = threadIdx.x / C8, npl = blockDim.x / C8;
    const int p0 = blockIdx.x * chunk, p1 = min(p0 + chunk, HW);
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    if (pl < npl) {
        for (int p = p0 + pl; p < p1; p += npl) {
            const f16x8 v = *(const f16x8 *)(x + ((int64_t)b * HW + p) * ldc + c8 * 8);
            if (lo_off && lo8_pa >= 0) {
                float l[8];
                lo8_load8((const char *)(x + ((int64_t)b * HW + p) * ldc) + 3 * lo_off + c8 * 8, __builtin_ldexpf(1.f, -(lo8_pa + 12)), l);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)v[j] + l[j]; s[j] += f; q[j] += f * f; }
            } else if (lo_off) {
                const f16x8 l = *(const f16x8 *)(x + ((int64_t)b * HW + p) * ldc + c8 * 8 + lo_off);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)v[j] + (float)l[j]; s[j] += f; q[j] += f * f; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; s[j] += f; q[j] += f * f; }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x * 16 + j] = s[j]; red[threadIdx.x * 16 + 8 + j] = q[j]; }
    __syncthreads();
    if (pl == 0) {
        for (int o = 1; o < npl; ++o)
#pragma unroll
            for (int j = 0; j < 16; ++j) red[c8 * 16 + j] += red[(o * C8 + c8) * 16 + j];
        float *dst = part + (((int64_t)blockIdx.x * gridDim.y + b) * C8 * 8 + c8 * 8) * 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) { dst[j * 2] = red[c8 * 16 + j]; dst[j * 2 + 1] = red[c8 * 16 + 8 + j]; }
    }
}

__global__ __launch_bounds__(256) void in_finalize_kernel(const float *__restrict__ part, int nchunk, int BC, float inv_hw,
                                                           float *__restrict__ stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BC) return;
    float s = 0.f, q = 0.f;
    for (int k = 0; k < nchunk; ++k) {
        const float2 v = *(const float2 *)(part + ((int64_t)k * BC + i) * 2);
        s += v.x; q += v.y;
    }
    const float mean = s * inv_hw;
    const float var = fmaxf(q * inv_hw - mean * mean, 0.f);
    stats[i * 2] = mean;
    stats[i * 2 + 1] = rsqrtf(var + 1e-5f);
}

// out = relu( relu(IN(a)) + (b ? (sb ? IN(b) : b) : 0) )   [second relu only when b is given];  sa / sb = {mean, rstd}
__global__ __launch_bounds__(256) void in_apply_kernel(const f16 *__restrict__ a, const float *__restrict__ sa,
                                                        const f16 *__restrict__ bsrc, const float *__restrict__ sb,
                                                        f16 *__restrict__ out, int B, int HW, int C8, int ldc, int lo_off, int lo8_pa) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * HW * C8) return;
    const int c8 = (int)(i % C8);
    const int64_t pix = i / C8;
    const int b = (int)(pix / HW);
    const int64_t o = pix * ldc + c8 * 8;
    const f16x8 va = *(const f16x8 *)(a + o);
    f16x8 vb;
    float la[8], lb[8];
    if (bsrc) vb = *(const f16x8 *)(bsrc + o);
    if (lo_off && lo8_pa >= 0) {    // split maps with e4m3 residual parts [hi | hi8 | lo8]
        const float inv = __builtin_ldexpf(1.f, -(lo8_pa + 12));
        lo8_load8((const char *)(a + pix * ldc) + 3 * lo_off + c8 * 8, inv, la);
        if (bsrc) lo8_load8((const char *)(bsrc + pix * ldc) + 3 * lo_off + c8 * 8, inv, lb);
    } else if (lo_off) {            // split-fp16 maps [hi | lo]
        const f16x8 ta = *(const f16x8 *)(a + o + lo_off);
#pragma unroll
        for (int j = 0; j < 8; ++j) la[j] = (float)ta[j];
        if (bsrc) {
            const f16x8 tb = *(const f16x8 *)(bsrc + o + lo_off);
#pragma unroll
            for (int j = 0; j < 8; ++j) lb[j] = (float)tb[j];
        }
    }
    const f32x4 *st = (const f32x4 *)(sa + ((int64_t)b * C8 * 8 + c8 * 8) * 2);
    f32x4 ms[4], ms2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ms[j] = st[j];
    if (bsrc && sb) {
        const f32x4 *s2 = (const f32x4 *)(sb + ((int64_t)b * C8 * 8 + c8 * 8) * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) ms2[j] = s2[j];
    }
    f16x8 r, rl;
    float rv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float mean = ms[j >> 1][(j & 1) * 2], rstd = ms[j >> 1][(j & 1) * 2 + 1];
        const float xa = lo_off ? (float)va[j] + la[j] : (float)va[j];
        float v = fmaxf((xa - mean) * rstd, 0.f);
        if (bsrc) {
            float w = lo_off ? (float)vb[j] + lb[j] : (float)vb[j];
            if (sb) w = (w - ms2[j >> 1][(j & 1) * 2]) * ms2[j >> 1][(j & 1) * 2 + 1];
            v = fmaxf(v + w, 0.f);
        }
        r[j] = (f16)v;
        rl[j] = (f16)(v - (float)r[j]);
        rv[j] = v;
    }
    *(f16x8 *)(out + o) = r;
    if (lo_off && lo8_pa >= 0) {
        const float shi = __builtin_ldexpf(1.f, lo8_pa), slo = __builtin_ldexpf(1.f, lo8_pa + 12);
        int2 h8, l8;
        h8.x = pb_fp8x4((float)r[0] * shi, (float)r[1] * shi, (float)r[2] * shi, (float)r[3] * shi);
        h8.y = pb_fp8x4((float)r[4] * shi, (float)r[5] * shi, (float)r[6] * shi, (float)r[7] * shi);
        l8.x = pb_fp8x4((rv[0] - (float)r[0]) * slo, (rv[1] - (float)r[1]) * slo, (rv[2] - (float)r[2]) * slo, (rv[3] - (float)r[3]) * slo);
        l8.y = pb_fp8x4((rv[4] - (float)r[4]) * slo, (rv[5] - (float)r[5]) * slo, (rv[6] - (float)r[6]) * slo, (rv[7] - (float)r[7]) * slo);
        char *pb = (char *)(out + pix * ldc);
        *(int2 *)(pb + 2 * lo_off + c8 * 8) = h8;
        *(int2 *)(pb + 3 * lo_off + c8 * 8) = l8;
    } else if (lo_off) *(f16x8 *)(out + o + lo_off) = rl;
}

// cnet output [rows][256] fp16 -> net = tanh(c[:128]) (fp32 master + fp16 copy in HX[:, 0:128]),
// inp = relu(c[128:]) in HX[:, 128:256]; flow = 0.
__global__ __launch_bounds__(256) void init_state_kernel(const f16 *__restrict__ c, float *__restrict__ h32,
                                                          f16 *__restrict__ hx, f16 *__restrict__ hx2, float *__restrict__ flow, int64_t rows,
                                                          int ld, int o8_off, float o8_scale, int inp_off) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 32) return;
    const int c8 = (int)(i % 32);
    const int64_t r = i / 32;
    const f16x8 v = *(const f16x8 *)(c + r * 256 + c8 * 8);
    f16x8 o;
    if (c8 < 16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = tanhf((float)v[j]);
            h32[r * 128 + c8 * 8 + j] = t;
            o[j] = (f16)t;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)fmaxf((float)v[j], 0.f);
    }
    const int dc = c8 < 16 ? c8 * 8 : inp_off + (c8 - 16) * 8;      // h at 0, the context features at inp_off (128, or 256 with the hoisted layout [h | motion | inp])
    *(f16x8 *)(hx + r * ld + dc) = o;
    if (c8 >= 16) *(f16x8 *)(hx2 + r * ld + dc) = o;          // the context part of the q conv's input
    if (o8_off) {
        int2 o8;
        o8.x = pb_fp8x4((float)o[0] * o8_scale, (float)o[1] * o8_scale, (float)o[2] * o8_scale, (float)o[3] * o8_scale);
        o8.y = pb_fp8x4((float)o[4] * o8_scale, (float)o[5] * o8_scale, (float)o[6] * o8_scale, (float)o[7] * o8_scale);
        *(int2 *)((char *)(hx + r * ld) + o8_off + dc) = o8;
        if (c8 >= 16) *(int2 *)((char *)(hx2 + r * ld) + o8_off + dc) = o8;
    }
    if (c8 == 0) { flow[r * 2] = 0.f; flow[r * 2 + 1] = 0.f; }
}


// ------------------------------------------------------------------------------------------------
// 9x9 x 4-level lookup (corr.py:29-50): channel k = l*81 + i*9 + j samples x = cx/2^l + (i-4),
// y = cy/2^l + (j-4) with bilinear weights in pixel coordinates and zeros outside the level.
// One thread per (pixel, level, i): the 9 j-samples share the x taps; output fp16 [rows][384].
// ------------------------------------------------------------------------------------------------
// avg_pool2d(2, 2) with floor on odd sizes, NHWC fp16, 8 channels per thread: the pooled target feature maps whose
// correlation with fmap1 IS the pooled correlation volume (corr.py:22-27; the pooling is linear)
__global__ __launch_bounds__(256) void avgpool2_nhwc_kernel(const f16 *__restrict__ x, f16 *__restrict__ y, int n, int H, int W, int OH,
                                                            int OW, int C8) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * OH * OW * C8) return;
    const int c = (int)(i % C8);
    const int64_t pix = i / C8;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), b = (int)(pix / ((int64_t)OW * OH));
    const f16 *base = x + (((int64_t)b * H + 2 * oy) * W + 2 * ox) * C8 * 8 + c * 8;
    const f16x8 v00 = *(const f16x8 *)base, v01 = *(const f16x8 *)(base + C8 * 8);
    const f16x8 v10 = *(const f16x8 *)(base + (int64_t)W * C8 * 8), v11 = *(const f16x8 *)(base + (int64_t)(W + 1) * C8 * 8);
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)(0.25f * (((float)v00[j] + (float)v01[j]) + ((float)v10[j] + (float)v11[j])));
    *(f16x8 *)(y + i * 8) = o;
}

// Target features of one pyramid level -> the B operand of the correlation GEMM: scaled by 1/16 (with fmap1 unscaled that
// is corr.py:58's 1/sqrt(256), exact in fp16) and re-ordered into 8 x 8 pixel tiles, row(y, x) = tile * 64 + (y & 7) * 8 +
// (x & 7), tile = (y >> 3) * (wp >> 3) + (x >> 3).  The GEMM then writes every source pixel's correlation row in that order,
// so the 10 x 10 window a lookup reads spans ~5 tiles of 128 bytes instead of 10 rows in 13 lines.  Pad rows stay zero.
__global__ __launch_bounds__(256) void corr_tile_kernel(const f16 *__restrict__ x, f16 *__restrict__ y, int F, int h, int w, int wp,
                                                         int npad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)F * h * w * 32) return;
    const int c = (int)(i & 31);
    const int64_t pix = i >> 5;
    const int px = (int)(pix % w), py = (int)((pix / w) % h), f = (int)(pix / ((int64_t)w * h));
    const f16x8 v = *(const f16x8 *)(x + pix * 256 + c * 8);
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)((float)v[j] * 0.0625f);
    const int row = ((py >> 3) * (wp >> 3) + (px >> 3)) * 64 + (py & 7) * 8 + (px & 7);
    *(f16x8 *)(y + ((int64_t)f * npad + row) * 256 + c * 8) = o;
}

struct PyrPtrs { const f16 *lv[4]; int h[4], w[4], wp[4], ld[4], hp8[4]; };
// a per-LANE level index into the kernel-argument arrays above compiles to loads from the kernarg segment (and a wait for every load in flight
// in front of each use); four-way selects on the scalar registers keep the level's geometry out of memory
// (as bit selects: nested ?: on the lane's level became exec-masked branches)
__device__ __forceinline__ unsigned sel4u(unsigned a0, unsigned a1, unsigned a2, unsigned a3, int l) {
    const unsigned m1 = 0u - (unsigned)(l & 1), m2 = 0u - (unsigned)((l >> 1) & 1);
    const unsigned lo = (a0 & ~m1) | (a1 & m1), hi = (a2 & ~m1) | (a3 & m1);
    return (lo & ~m2) | (hi & m2);
}
__device__ __forceinline__ int sel4(const int (&a)[4], int l) { return (int)sel4u((unsigned)a[0], (unsigned)a[1], (unsigned)a[2], (unsigned)a[3], l); }
__device__ __forceinline__ const f16 *sel4(const f16 *const (&a)[4], int l) {
    const unsigned long long p0 = (unsigned long long)a[0], p1 = (unsigned long long)a[1], p2 = (unsigned long long)a[2], p3 = (unsigned long long)a[3];
    const unsigned lo = sel4u((unsigned)p0, (unsigned)p1, (unsigned)p2, (unsigned)p3, l);
    const unsigned hi = sel4u((unsigned)(p0 >> 32), (unsigned)(p1 >> 32), (unsigned)(p2 >> 32), (unsigned)(p3 >> 32), l);
    return (const f16 *)(((unsigned long long)hi << 32) | lo);
}      // hp8 = ld / wp: padded target rows of the level

// CorrBlock.__call__ (corr.py:29-50): 9 x 9 bilinear window on each of the 4 levels around coords / 2^l, zero outside.
// A block covers 7 pixels, i.e. 28 (pixel, level) windows:
//   1. the 9 + 9 sample coordinates of every window, once (the reference's normalise / un-normalise round trip costs a
//      division each);
//   2. the window itself, 11 rows x 3 tile-row segments of 8 targets, as aligned 16-byte loads from the tiled volume into
//      the LDS - out-of-range rows / tiles are stored as zeros, which is the reference's padding_mode='zeros';
//   3. one thread per (pixel, level, window column) blends its 9 rows from the LDS copy (no bounds tests left);
//   4. the 7 x 324 results leave as whole 8-byte vectors of consecutive channels.
// (The scalar-gather version of step 3 - 36 two-byte global loads per thread - took 215 us per call at 720p x 8 pairs
// whatever the volume layout; it was bound by the number of gather instructions.)
__global__ __launch_bounds__(256) void corr_lookup_kernel(PyrPtrs py, const float *__restrict__ flow, int P, int w8,
                                                           f16 *__restrict__ out, int64_t rows, int ldo, int o8_off, float o8_scale) {
    __shared__ __attribute__((aligned(16))) f16 win[28 * 11 * 24];
    __shared__ __attribute__((aligned(8))) f16 tile[7 * 324];
    __shared__ int ci[28 * 18];
    __shared__ float ca[28 * 18];
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 7;
    {   // two coordinates per thread; both flow loads are issued before either is used (one round trip instead of two to four)
        f32x2 fl[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = t + 256 * u, idc = idx < 28 * 18 ? idx : 0;
            const int pr = (idc / 18) >> 2;
            const int64_t r = r0 + pr < rows ? r0 + pr : rows - 1;
            fl[u] = *(const f32x2 *)(flow + r * 2);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = t + 256 * u, idc = idx < 28 * 18 ? idx : 0;
            const int k = idc % 18, pl = idc / 18, l = pl & 3, pr = pl >> 2;
            const int64_t r = r0 + pr < rows ? r0 + pr : rows - 1;
            const int p = (int)((unsigned)r % (unsigned)P);           // rows < 2^31 (launcher check): a 32-bit remainder, not the 64-bit library one
            const float inv = 1.f / (float)(1 << l);
            const bool isx = k < 9;
            // x samples k = 0..8 around column p % w8, y samples k = 9..17 around row p / w8; the same expression for both
            const float c = (float)(isx ? p % w8 : p / w8) + (isx ? fl[u][0] : fl[u][1]);
            const int dim = isx ? sel4(py.w, l) : sel4(py.h, l);
            // reproduce grid_sample's round trip: normalise then un-normalise (align_corners=True)
            const float v = ((2.f * (c * inv + (float)(isx ? k - 4 : k - 13)) / (float)(dim - 1) - 1.f) + 1.f) * 0.5f * (float)(dim - 1);
            const float f = floorf(v);
            if (idx < 28 * 18) {
                // far out-of-range samples are all zero anyway: clamp so that the integer arithmetic below cannot overflow
                ci[idx] = (int)fminf(fmaxf(f, -65536.f), 65536.f);
                ca[idx] = v - f;
            }
        }
    }
    __syncthreads();
    // all of a thread's (up to four) window loads are issued before the first one is consumed: as a plain loop every iteration waited for its
    // own load in front of its LDS store - four HBM round trips in a row per block (round 4: the kernel moved 1.3 TB/s with 24 waves per CU)
    {
        f16x8 wv[4];
        int wdst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = t + 256 * u;
            const int idc = idx < 28 * 33 ? idx : 0;
            const int seg = idc % 3, row = (idc / 3) % 11, pl = idc / 33, l = pl & 3, pr = pl >> 2;
            const int64_t r = r0 + pr < rows ? r0 + pr : rows - 1;
            const int hp8 = sel4(py.hp8, l), wt = sel4(py.wp, l) >> 3;
            const int y = ci[pl * 18 + 9] + row, tx = (ci[pl * 18] >> 3) + seg;
            // the third segment is only read by the blend when the window's last column + 1 reaches it, i.e. when the first column sits on the
            // last target of its tile (1 window in 8): skipping it otherwise takes a quarter of the kernel's fetches away
            const bool need = seg < 2 || ci[pl * 18 + 8] + 1 - (ci[pl * 18] & ~7) >= 16;
            const bool ok = idx < 28 * 33 && need && (unsigned)y < (unsigned)hp8 && (unsigned)tx < (unsigned)wt;
            // branch-free: an out-of-range window row reads the level's first 16 bytes and is zeroed afterwards (zeros = padding_mode='zeros')
            const f16 *src = sel4(py.lv, l) + (ok ? r * (int64_t)sel4(py.ld, l) + ((y >> 3) * wt + tx) * 64 + (y & 7) * 8 : 0);
            const f16x8 ld = *(const __attribute__((address_space(1))) f16x8 *)(unsigned long long)src;      // (a global, not a flat load: the pointer went through integers)
            const f16x8 z = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
            wv[u] = ok ? ld : z;
            wdst[u] = idx < 28 * 33 ? (pl * 11 + row) * 24 + seg * 8 : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (wdst[u] >= 0) *(f16x8 *)(win + wdst[u]) = wv[u];
    }
    __syncthreads();
    if (t < 252) {
        const int li = t % 36, pr = t / 36;
        const int l = li / 9, wi = li - l * 9;
        const int pl = pr * 4 + l, cb = pl * 18;
        const int xb = ci[cb] & ~7, yb = ci[cb + 9];
        int xo = ci[cb + wi] - xb;
        const float ax = ca[cb + wi];
        xo = xo < 0 ? 0 : (xo > 22 ? 22 : xo);                  // always in range (spread of the 9 floors <= 9); defensive
        const f16 *wp = win + pl * 11 * 24 + xo;
        f16 *dst = tile + pr * 324 + l * 81 + wi * 9;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            int yo = ci[cb + 9 + j] - yb;
            const float ay = ca[cb + 9 + j];
            yo = yo < 0 ? 0 : (yo > 9 ? 9 : yo);
            const float v00 = (float)wp[yo * 24], v01 = (float)wp[yo * 24 + 1];
            const float v10 = (float)wp[yo * 24 + 24], v11 = (float)wp[yo * 24 + 25];
            dst[j] = (f16)(v00 * (1.f - ax) * (1.f - ay) + v01 * ax * (1.f - ay) + v10 * (1.f - ax) * ay + v11 * ax * ay);
        }
    }
    __syncthreads();
    for (int v = t; v < 7 * 81; v += 256) {
        const int pr = v / 81, c4 = v - pr * 81;
        const int64_t r = r0 + pr;
        if (r < rows) {
            const f16x4 o = *(const f16x4 *)(tile + pr * 324 + c4 * 4);
            *(f16x4 *)(out + r * ldo + c4 * 4) = o;
            if (o8_off)      // fp8 copy after the row's fp16 part: the A operand of convc1's MX segment (gemm.h nk16)
                *(int *)((char *)(out + r * ldo) + o8_off + c4 * 4) = pb_fp8x4((float)o[0] * o8_scale, (float)o[1] * o8_scale, (float)o[2] * o8_scale, (float)o[3] * o8_scale);
        }
    }
}

// flow (fp32 [rows][2]) -> HX[:, 382:384] (the last two input channels of the GRU, update.py:97)
__global__ void put_flow_kernel(const float *__restrict__ flow, f16 *__restrict__ hx, f16 *__restrict__ hx2, int64_t rows, int ld, int o8_off,
                                float o8_scale, int flow_off) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    f16x2 o;
    o[0] = (f16)flow[r * 2]; o[1] = (f16)flow[r * 2 + 1];
    *(f16x2 *)(hx + r * ld + flow_off) = o;
    *(f16x2 *)(hx2 + r * ld + flow_off) = o;
    if (o8_off) {
        const unsigned short o8 = pb_fp8x2((float)o[0] * o8_scale, (float)o[1] * o8_scale);
        *(unsigned short *)((char *)(hx + r * ld) + o8_off + flow_off) = o8;
        *(unsigned short *)((char *)(hx2 + r * ld) + o8_off + flow_off) = o8;
    }
}

// r * h and h = (1 - z) h + z q are epilogues of the z / r and q convolutions (gemm.h ACT_GRU_ZR / ACT_GRU_Q)

// FlowHead conv2 (update.py:11-12: 3x3, 256 -> 2) + coords1 += delta_flow, as a direct convolution: with two output
// channels an implicit GEMM spends 98 % of its MFMAs on padding columns (73 us per iteration at 720p / 8 pairs; this: 36 us).
// Half a wave per run of 16 pixels, 8 input channels per lane; the 2 x 9 x 8 weights of a lane and a sliding 3 x 3 window of
// channel vectors stay in registers, products accumulate in fp32 (v_dot2_f32_f16), one 5-step butterfly per pixel.
// SPLIT: the packed weights are [o][tap][w_hi (256) | w_lo (256)] (split-fp16 precision mode) and both parts are applied.
template <bool SPLIT>
__global__ __launch_bounds__(256) void flow_head2_kernel(const f16 *__restrict__ x, const f16 *__restrict__ w, const float *__restrict__ bias,
                                                          float *__restrict__ flow, int nseg, int H, int W, int segw) {
    // half a wave walks a horizontal run of `segw` pixels with a 3 x 3 window of channel vectors in registers: every step
    // loads one new column (3 vectors) instead of 9
    const int lane = threadIdx.x & 63, half = lane >> 5, cl = lane & 31;
    constexpr int WS = SPLIT ? 512 : 256;
    f16x8 wv[2][9], wl[SPLIT ? 2 : 1][SPLIT ? 9 : 1];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            wv[o][t] = *(const f16x8 *)(w + (o * 9 + t) * WS + cl * 8);
            if constexpr (SPLIT) wl[o][t] = *(const f16x8 *)(w + (o * 9 + t) * WS + 256 + cl * 8);
        }
    const float b0 = bias[0], b1 = bias[1];
    const int wid = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int seg = wid * 2 + half;                    // segment = (image, row, run of segw pixels)
    if (seg >= nseg) return;
    const int spr = (W + segw - 1) / segw;             // segments per row
    const int sx = seg % spr, y = (seg / spr) % H, img = seg / (spr * H);
    const int xa = sx * segw, xb = xa + segw < W ? xa + segw : W;
    const f16 *base = x + ((int64_t)img * H * W) * 256 + cl * 8;
    f16x8 zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero[j] = (f16)0.f;
    auto col = [&](int ix, f16x8 (&c)[3]) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + ky - 1;
            c[ky] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? *(const f16x8 *)(base + ((int64_t)iy * W + ix) * 256) : zero;
        }
    };
    // the window's columns are loaded TWO steps ahead (round 5: at 202 registers the kernel runs two waves per SIMD, so a column requested at the
    // top of the step that needs it could expose an L2 round trip per pixel).  Measured: 3.28 -> 3.18 ms per step - the kernel is bound by its
    // two 72-deep v_dot2 chains per pixel pair, not by that latency
    f16x8 c0[3], c1[3], c2[3], c3[3];
    col(xa - 1, c0); col(xa, c1); col(xa + 1, c2);
    for (int ix = xa; ix < xb; ++ix) {
        col(ix + 2, c3);
        // the pixel's current flow is requested here, with the window column, instead of as a dependent load in front of the store.
        // (Round 5 went through this kernel's suspects - this round trip, column prefetch depth, eight instead of two accumulation chains,
        // XCD-contiguous block order, run lengths 8 ... 90: 3.06-3.38 ms per step whatever is changed (profiles/r05r_flow_head2_sweep.txt);
        // what is left is the issue rate of its 144 v_dot2 per pixel pair.)
        const int64_t p = ((int64_t)img * H + y) * W + ix;
        const f32x2 old = *(const f32x2 *)(flow + p * 2);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f16x8 v = kx == 0 ? c0[ky] : (kx == 1 ? c1[ky] : c2[ky]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f16x2 vv = {v[2 * j], v[2 * j + 1]};
                    const f16x2 w0 = {wv[0][ky * 3 + kx][2 * j], wv[0][ky * 3 + kx][2 * j + 1]};
                    const f16x2 w1 = {wv[1][ky * 3 + kx][2 * j], wv[1][ky * 3 + kx][2 * j + 1]};
                    a0 = __builtin_amdgcn_fdot2(vv, w0, a0, false);
                    a1 = __builtin_amdgcn_fdot2(vv, w1, a1, false);
                    if constexpr (SPLIT) {
                        const f16x2 l0 = {wl[0][ky * 3 + kx][2 * j], wl[0][ky * 3 + kx][2 * j + 1]};
                        const f16x2 l1 = {wl[1][ky * 3 + kx][2 * j], wl[1][ky * 3 + kx][2 * j + 1]};
                        a0 = __builtin_amdgcn_fdot2(vv, l0, a0, false);
                        a1 = __builtin_amdgcn_fdot2(vv, l1, a1, false);
                    }
                }
            }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); }
        if (cl == 0) {
            f32x2 o;
            o[0] = old[0] + (a0 + b0); o[1] = old[1] + (a1 + b1);
            *(f32x2 *)(flow + p * 2) = o;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) { c0[ky] = c1[ky]; c1[ky] = c2[ky]; c2[ky] = c3[ky]; }
    }
}

// ------------------------------------------------------------------------------------------------
// convex upsample (raft.py:73-84) fused with the unpad and the per-flow max displacement:
// mask fp32 [rows][576] (already * 0.25).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f2ord_(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// one wave per 1/8-resolution pixel, lane = sub-pixel (sy * 8 + sx): the nine 64-wide mask groups are coalesced reads
__global__ __launch_bounds__(256) void upsample_kernel(const float *__restrict__ flow, const float *__restrict__ mask, int h8,
                                                        int w8, int pad_l, int pad_t, int sh, int sw,
                                                        float *__restrict__ out, unsigned *__restrict__ maxd) {
    const int n = blockIdx.y;
    const int P = h8 * w8;
    float dmax = 0.f;
    const int sub = threadIdx.x & 63;
    // grid-stride over the low-res pixels so that a wave issues ONE atomicMax at the end (same-address atomics serialise)
    for (int p = blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += gridDim.x * 4) {
        const int py = p / w8, px = p - py * w8;
        const float *m = mask + ((int64_t)n * P + p) * 576 + sub;
        float e[9], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) { e[k] = m[k * 64]; mx = fmaxf(mx, e[k]); }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { e[k] = __expf(e[k] - mx); den += e[k]; }
        float u = 0.f, v = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = py + k / 3 - 1, xx = px + k % 3 - 1;
            if ((unsigned)yy < (unsigned)h8 && (unsigned)xx < (unsigned)w8) {
                const float *f = flow + ((int64_t)n * P + yy * w8 + xx) * 2;
                const float wgt = e[k] / den;
                u += wgt * (8.f * f[0]);
                v += wgt * (8.f * f[1]);
            }
        }
        const int oy = py * 8 + (sub >> 3) - pad_t, ox = px * 8 + (sub & 7) - pad_l;
        if ((unsigned)oy < (unsigned)sh && (unsigned)ox < (unsigned)sw) {
            float *o = out + (((int64_t)n * sh + oy) * sw + ox) * 2;
            o[0] = u; o[1] = v;
            dmax = fmaxf(dmax, ex_fsqrt(ex_fadd(ex_fmul(u, u), ex_fmul(v, v))));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o));
    if ((threadIdx.x & 63) == 0) atomicMax(maxd + n, f2ord_(dmax));
}

// process_flow (encode.py:98-126): float32 distance / angle, float64 colour ramp + saturation blend, truncation.
// atan2 in double from separately rounded IEEE operations only (no library call, no FMA contraction), so that the CPU oracle
// (oracle/raft_oracle.py atan2_rn, the same operation sequence in numpy float64) reproduces it bit for bit.  t = min / max of
// the magnitudes in [0, 1]; nearest c in {0, 1/4, 1/2, 3/4, 1}; u = (t - c) / (1 + t c), |u| <= 1/8; atan(t) = atan(c) + the
// odd Taylor series of atan(u) through u^17 (truncation < 8e-19); then the octant fix-ups.  |error| < 5e-16, i.e. the float32
// rounding of the result is the correctly rounded float32 arctan2 (0 mismatches in 2e6 random inputs against a long double
// evaluation; tests/test_raft_oracle.py).
__device__ __forceinline__ double atan2_rn(double y, double x) {
    const double ATAN_C[5] = {0.0, 0.24497866312686414, 0.4636476090008061, 0.6435011087932844, 0.7853981633974483};
    const double ax = fabs(x), ay = fabs(y);
    const double hi = fmax(ax, ay), lo = fmin(ax, ay);
    const double t = hi == 0.0 ? 0.0 : ex_ddiv(lo, hi);
    if (t != t) return t;                                   // NaN in -> NaN out (0 / 0 flow: the last frame of a video)
    const double k = floor(ex_dadd(ex_dmul(t, 4.0), 0.5));
    const double c = ex_dmul(k, 0.25);
    const double u = ex_ddiv(ex_dsub(t, c), ex_dadd(1.0, ex_dmul(t, c)));
    const double u2 = ex_dmul(u, u);
    double p = 1.0 / 17.0;
    p = ex_dadd(ex_dmul(p, u2), -1.0 / 15.0);
    p = ex_dadd(ex_dmul(p, u2), 1.0 / 13.0);
    p = ex_dadd(ex_dmul(p, u2), -1.0 / 11.0);
    p = ex_dadd(ex_dmul(p, u2), 1.0 / 9.0);
    p = ex_dadd(ex_dmul(p, u2), -1.0 / 7.0);
    p = ex_dadd(ex_dmul(p, u2), 1.0 / 5.0);
    p = ex_dadd(ex_dmul(p, u2), -1.0 / 3.0);
    p = ex_dadd(ex_dmul(p, u2), 1.0);
    double r = ex_dadd(ATAN_C[(int)k], ex_dmul(u, p));
    if (ay > ax) r = ex_dsub(1.5707963267948966, r);
    if (x < 0.0) r = ex_dsub(3.141592653589793, r);
    if (y < 0.0) r = -r;
    return r;
}

__global__ __launch_bounds__(256) void flow_encode_kernel(const float *__restrict__ flow, int64_t per,
                                                           const unsigned *__restrict__ maxd, uint8_t *__restrict__ rgb,
                                                           float *__restrict__ max_out) {
    const int n = blockIdx.y;
    const unsigned mu = maxd[n];
    const float mx = __uint_as_float((mu & 0x80000000u) ? (mu & 0x7fffffffu) : ~mu);
    if (blockIdx.x == 0 && threadIdx.x == 0 && max_out) max_out[n] = mx;
    if (!rgb) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
        const float *f = flow + ((int64_t)n * per + i) * 2;
        const float dx = ex_fdiv(f[0], mx), dy = ex_fdiv(f[1], mx);
        const float rad = ex_fsqrt(ex_fadd(ex_fmul(dx, dx), ex_fmul(dy, dy)));
        // np.arctan2 on float32 is not one function: numpy dispatches to SVML (<= 4 ULP) on AVX512 hosts and to libm elsewhere, so
        // the reference's last bit depends on its CPU.  The engine (and oracle.process_flow(exact_atan2=True)) take the
        // correctly rounded float32 value, from a double evaluation made of IEEE +, -, *, / only (atan2_rn).
        const float at = (float)atan2_rn((double)dy, (double)dx);
        const float a = ex_fmul(ex_fadd(ex_fdiv(at, 3.14159265358979323846f), 1.0f), 0.5f);
        const float h6 = ex_fmul(a, 6.0f);
        const float offs[3] = {0.f, 4.f, 2.f};
        uint8_t o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = (double)(c == 0 ? h6 : ex_fadd(h6, offs[c]));
            v = fmod(v, 6.0);
            if (v < 0.0) v = ex_dadd(v, 6.0);
            v = ex_dsub(fabs(ex_dsub(v, 3.0)), 1.0);
            v = fmin(fmax(v, 0.0), 1.0);
            v = ex_dadd(ex_dmul(v, (double)rad), (double)ex_fsub(1.0f, rad));
            v = ex_dmul(v, 255.0);
            o[c] = (v == v) ? (uint8_t)(int)v : (uint8_t)0;
        }
        uint8_t *d = rgb + ((int64_t)n * per + i) * 3;
        d[0] = o[0]; d[1] = o[1]; d[2] = o[2];
    }
}

__global__ void fill_u32_kernel(unsigned *p, unsigned v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

#define LAUNCH_CHECK() do { PB_HIP(hipGetLastError()); return 0; } while (0)

int launch_raft_prep(hipStream_t s, const uint8_t *frames, int F, int H, int W, int sh, int sw, int Hp, int Wp, int pad_l,
                     int pad_t, int resize, const int *xi, const int *xc, const int *yi, const int *yc, f16 *out,
                     uint8_t *scaled_out, int s2d, int lo_off, int lo8_pa, int norm_mode, int isz) {
    hipLaunchKernelGGL(raft_prep_kernel, dim3(nblk((int64_t)F * Hp * Wp)), dim3(256), 0, s, frames, F, H, W, sh, sw, Hp, Wp,
                       pad_l, pad_t, resize, xi, xc, yi, yc, out, scaled_out, s2d, lo_off, lo8_pa, norm_mode, isz);
    LAUNCH_CHECK();
}
int launch_im2col7_flow(hipStream_t s, const float *x, int B, int H, int W, f16 *out, int Kp, int ld, int o8) {
    hipLaunchKernelGGL((im2col7_kernel<float, 2>), dim3(nblk((int64_t)B * H * W * 49)), dim3(256), 0, s, x, B, H, W, 2, 1, H, W,
                       out, Kp, ld ? ld : Kp, o8);
    LAUNCH_CHECK();
}
int in_stats_chunks(int HW) { return (HW + 2047) / 2048; }
int launch_in_stats(hipStream_t s, const f16 *x, int B, int HW, int C, int ldc, float *part, float *stats, int lo_off, int lo8_pa) {
    PB_CHECK(C % 8 == 0 && C <= 256 && 256 % (C / 8) == 0, -1, "instance norm: C=%d unsupported", C);
    const int chunk = 2048, nchunk = in_stats_chunks(HW);
    hipLaunchKernelGGL(in_stats_kernel, dim3(nchunk, B), dim3(256), 0, s, x, HW, C / 8, ldc, part, chunk, lo_off, lo8_pa);
    hipLaunchKernelGGL(in_finalize_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, part, nchunk, B * C, 1.f / (float)HW, stats);
    LAUNCH_CHECK();
}
int launch_in_apply(hipStream_t s, const f16 *a, const float *sa, const f16 *b, const float *sb, f16 *out, int B, int HW,
                    int C, int ldc, int lo_off, int lo8_pa) {
    hipLaunchKernelGGL(in_apply_kernel, dim3(nblk((int64_t)B * HW * (C / 8))), dim3(256), 0, s, a, sa, b, sb, out, B, HW, C / 8, ldc, lo_off, lo8_pa);
    LAUNCH_CHECK();
}
int launch_init_state(hipStream_t s, const f16 *c, float *h32, f16 *hx, f16 *hx2, float *flow, int64_t rows, int ld, int o8_off, float o8_scale, int inp_off) {
    hipLaunchKernelGGL(init_state_kernel, dim3(nblk(rows * 32)), dim3(256), 0, s, c, h32, hx, hx2, flow, rows, ld, o8_off, o8_scale, inp_off);
    LAUNCH_CHECK();
}
int launch_avgpool2_nhwc(hipStream_t s, const f16 *x, f16 *y, int n, int H, int W, int C) {
    const int OH = H / 2, OW = W / 2;
    hipLaunchKernelGGL(avgpool2_nhwc_kernel, dim3(nblk((int64_t)n * OH * OW * (C / 8))), dim3(256), 0, s, x, y, n, H, W, OH, OW, C / 8);
    LAUNCH_CHECK();
}
int launch_corr_tile(hipStream_t s, const f16 *x, f16 *y, int F, int h, int w, int wp, int npad) {
    hipLaunchKernelGGL(corr_tile_kernel, dim3(nblk((int64_t)F * h * w * 32)), dim3(256), 0, s, x, y, F, h, w, wp, npad);
    LAUNCH_CHECK();
}
int launch_corr_lookup(hipStream_t s, const f16 *const lv[4], const int h[4], const int w[4], const int wp[4], const int ld[4],
                       const float *flow, int P, int w8, f16 *out, int64_t rows, int ldo, int o8_off, float o8_scale) {
    PyrPtrs py;
    for (int i = 0; i < 4; ++i) { py.lv[i] = lv[i]; py.h[i] = h[i]; py.w[i] = w[i]; py.wp[i] = wp[i]; py.ld[i] = ld[i]; py.hp8[i] = ld[i] / wp[i]; }
    PB_CHECK(rows < (1LL << 31), -1, "corr_lookup: %lld rows", (long long)rows);
    hipLaunchKernelGGL(corr_lookup_kernel, dim3((unsigned)((rows + 6) / 7)), dim3(256), 0, s, py, flow, P, w8, out, rows, ldo, o8_off, o8_scale);
    LAUNCH_CHECK();
}
// BasicMotionEncoder.convf1 (update.py:88: 7 x 7, 2 -> 128, ReLU) as a direct kernel on the fp32 flow field.  As an im2col GEMM it cost a 130 MB
// [rows, 128] operand written and read back per iteration for 14 GFLOP (im2col7 0.19 ms + GEMM 0.10 ms per launch at 31 pairs 1080p x 0.75).
// Here a wave owns 32 consecutive pixels: the A fragments (k = tap * 2 + channel, 98 -> 112 = 7 k-steps) are gathered straight from the flow
// field - 4.5 MB, L2 resident; lane (li, lh) needs taps 8 ks + 4 lh .. + 3 of pixel li, i.e. four 8-byte loads per k-step - and rounded to fp16
// exactly as the GEMM path's operand was; the weights [w_hi | w_lo] (fp16 pair, `passes` = 2 in the split mode) sit in the LDS for the
// kernel's lifetime, rows interleaved (LDS row g * 64 + tn * 32 + li <-> channel g * 64 + 2 li + tn) so a lane owns two adjacent channels and half
// a wave stores one contiguous 128-byte line per pixel.  Output as the GEMM epilogue's: relu(v + bias) as fp16 (+ its e4m3 copy at byte o8_off).
constexpr int F1_LD = 120;                     // halfs per weight row in the LDS (240 bytes: 16-byte fragment reads of 16 lanes hit 64 distinct banks)
__global__ __launch_bounds__(256, 2) void convf1_kernel(const float *__restrict__ flow, const f16 *__restrict__ wpk, const float *__restrict__ bias,
                                                         f16 *__restrict__ out, int64_t rows, int P, int h8, int w8, int ldo, int o8_off, float o8_scale,
                                                         int passes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int nw16 = passes * 128 * F1_LD / 8;                         // 16-byte chunks of the packed weights
    for (int i = tid; i < nw16; i += 256) ((f16x8 *)smem)[i] = ((const f16x8 *)wpk)[i];
    __syncthreads();
    // taps of this lane's fragment chunk: t = 8 ks + 4 lh + e -> (ky, kx) offsets relative to the pixel, 49.. = padding
    float b2[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g) { b2[g][0] = bias[g * 64 + 2 * li]; b2[g][1] = bias[g * 64 + 2 * li + 1]; }
    const int64_t ntiles = (rows + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (tid >> 6); tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t m = tile * 32 + li;
        const int64_t mc = m < rows ? m : rows - 1;
        const int64_t n = mc / P;
        const int rem = (int)(mc - n * P), y = rem / w8, x = rem - y * w8;
        const float *fb = flow + n * P * 2;
        f16x8 a[7];
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            f32x2 v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = 8 * ks + 4 * lh + e;
                const int ky = (t * 37) >> 8, kx = t - 7 * ky;          // t / 7 for t < 64
                const int iy = y + ky - 3, ix = x + kx - 3;
                const bool ok = t < 49 && (unsigned)iy < (unsigned)h8 && (unsigned)ix < (unsigned)w8;
                const f32x2 z = {0.f, 0.f};
                v[e] = ok ? *(const f32x2 *)(fb + ((int64_t)iy * w8 + ix) * 2) : z;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[ks][2 * e] = (f16)v[e][0]; a[ks][2 * e + 1] = (f16)v[e][1]; }
        }
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int ps = 0; ps < passes; ++ps) {
            const char *wb = smem + ps * 128 * F1_LD * 2 + li * (F1_LD * 2) + lh * 16;
#pragma unroll
            for (int ks = 0; ks < 7; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f16x8 b = *(const f16x8 *)(wb + j * 32 * (F1_LD * 2) + ks * 32);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b, acc[j], 0, 0, 0);
                }
        }
        // acc[g * 2 + tn][r]: pixel (r & 3) + 8 (r >> 2) + 4 lh of the tile, channel g * 64 + 2 li + tn
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t mr = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (mr < rows) {
                f16 *row = out + mr * ldo;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    f16x2 o;
                    o[0] = (f16)fmaxf(acc[g * 2][r] + b2[g][0], 0.f); o[1] = (f16)fmaxf(acc[g * 2 + 1][r] + b2[g][1], 0.f);
                    *(f16x2 *)(row + g * 64 + 2 * li) = o;
                    if (o8_off) *(unsigned short *)((char *)row + o8_off + g * 64 + 2 * li) = pb_fp8x2((float)o[0] * o8_scale, (float)o[1] * o8_scale);
                }
            }
        }
    }
}

int convf1_packed_halfs(int passes) { return passes * 128 * F1_LD; }
// host side of the layout above: w [128][2][7][7] fp32 -> [pass][LDS row][F1_LD] fp16 (pass 0: hi, pass 1: the fp16 rounding residual)
void convf1_pack(const float *w, int passes, f16 *dst) {
    for (int i = 0; i < passes * 128 * F1_LD; ++i) dst[i] = (f16)0.f;
    for (int row = 0; row < 128; ++row) {
        const int g = row >> 6, tn = (row >> 5) & 1, l = row & 31, ch = g * 64 + 2 * l + tn;
        for (int c = 0; c < 2; ++c)
            for (int tp = 0; tp < 49; ++tp) {
                const float v = w[((size_t)ch * 2 + c) * 49 + tp];
                const f16 hi = (f16)v;
                dst[(size_t)row * F1_LD + tp * 2 + c] = hi;
                if (passes > 1) dst[(size_t)(128 + row) * F1_LD + tp * 2 + c] = (f16)(v - (float)hi);
            }
    }
}
int launch_convf1(hipStream_t s, const float *flow, const f16 *wpk, const float *bias, f16 *out, int64_t rows, int P, int h8, int w8, int ldo,
                  int o8_off, float o8_scale, int passes) {
    const int smem = passes * 128 * F1_LD * 2;
    static bool attr = false;
    if (!attr) { PB_HIP(hipFuncSetAttribute((const void *)convf1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * F1_LD * 2)); attr = true; }
    const int64_t ntiles = (rows + 31) / 32;
    const int grid = (int)std::min<int64_t>((ntiles + 3) / 4, 512);
    hipLaunchKernelGGL(convf1_kernel, dim3(grid), dim3(256), smem, s, flow, wpk, bias, out, rows, P, h8, w8, ldo, o8_off, o8_scale, passes);
    LAUNCH_CHECK();
}
int launch_put_flow(hipStream_t s, const float *flow, f16 *hx, f16 *hx2, int64_t rows, int ld, int o8_off, float o8_scale, int flow_off) {
    hipLaunchKernelGGL(put_flow_kernel, dim3(nblk(rows)), dim3(256), 0, s, flow, hx, hx2, rows, ld, o8_off, o8_scale, flow_off);
    LAUNCH_CHECK();
}
int launch_flow_head2(hipStream_t s, const f16 *x, const f16 *w, const float *bias, float *flow, int n, int H, int W, int split) {
    const int segw = 16, nseg = n * H * ((W + segw - 1) / segw);
    if (split) hipLaunchKernelGGL(flow_head2_kernel<true>, dim3((nseg + 7) / 8), dim3(256), 0, s, x, w, bias, flow, nseg, H, W, segw);
    else hipLaunchKernelGGL(flow_head2_kernel<false>, dim3((nseg + 7) / 8), dim3(256), 0, s, x, w, bias, flow, nseg, H, W, segw);
    LAUNCH_CHECK();
}
int launch_upsample(hipStream_t s, const float *flow, const float *mask, int N, int h8, int w8, int pad_l, int pad_t, int sh,
                    int sw, float *out, unsigned *maxd) {
    hipLaunchKernelGGL(fill_u32_kernel, dim3(nblk(N)), dim3(256), 0, s, maxd, 0u, N);
    hipLaunchKernelGGL(upsample_kernel, dim3(std::min(nblk((int64_t)h8 * w8, 4), 256u), N), dim3(256), 0, s, flow, mask, h8, w8, pad_l, pad_t, sh,
                       sw, out, maxd);
    LAUNCH_CHECK();
}
// flow_gmflow --inference_size, the way back (reference flow_gmflow.py:92-97): F.interpolate(bilinear, align_corners = True) of the flow from the
// inference grid (ih, iw) to the scaled frame (sh, sw), u * sw / iw and v * sh / ih; the maximum displacement the encode needs is taken here
__global__ __launch_bounds__(256) void flow_resize_back_kernel(const float *__restrict__ in, int ih, int iw, int sh, int sw,
                                                               float *__restrict__ out, unsigned *__restrict__ maxd) {
    const int n = blockIdx.y;
    const float ry = sh > 1 ? (float)(ih - 1) / (float)(sh - 1) : 0.f, rx = sw > 1 ? (float)(iw - 1) / (float)(sw - 1) : 0.f;
    float dmax = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)sh * sw; i += (int64_t)gridDim.x * 256) {
        const int y = (int)(i / sw), x = (int)(i - (int64_t)y * sw);
        const float fy = ry * (float)y, fx = rx * (float)x;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < ih - 1 ? 1 : 0), x1 = x0 + (x0 < iw - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float *b = in + (int64_t)n * ih * iw * 2;
        float uv[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float p00 = b[((int64_t)y0 * iw + x0) * 2 + c], p01 = b[((int64_t)y0 * iw + x1) * 2 + c];
            const float p10 = b[((int64_t)y1 * iw + x0) * 2 + c], p11 = b[((int64_t)y1 * iw + x1) * 2 + c];
            const float v = (1.f - ly) * ((1.f - lx) * p00 + lx * p01) + ly * ((1.f - lx) * p10 + lx * p11);
            uv[c] = c == 0 ? ex_fdiv(ex_fmul(v, (float)sw), (float)iw) : ex_fdiv(ex_fmul(v, (float)sh), (float)ih);
        }
        float *o = out + ((int64_t)n * sh * sw + i) * 2;
        o[0] = uv[0]; o[1] = uv[1];
        dmax = fmaxf(dmax, ex_fsqrt(ex_fadd(ex_fmul(uv[0], uv[0]), ex_fmul(uv[1], uv[1]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o));
    if ((threadIdx.x & 63) == 0) atomicMax(maxd + n, f2ord_(dmax));
}
int launch_flow_resize_back(hipStream_t s, const float *in, int N, int ih, int iw, int sh, int sw, float *out, unsigned *maxd) {
    hipLaunchKernelGGL(fill_u32_kernel, dim3(nblk(N)), dim3(256), 0, s, maxd, 0u, N);
    hipLaunchKernelGGL(flow_resize_back_kernel, dim3(std::min(nblk((int64_t)sh * sw), 512u), N), dim3(256), 0, s, in, ih, iw, sh, sw, out, maxd);
    LAUNCH_CHECK();
}
int launch_flow_encode(hipStream_t s, const float *flow, int N, int sh, int sw, const unsigned *maxd, uint8_t *rgb,
                       float *max_out) {
    unsigned gx = nblk((int64_t)sh * sw);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(flow_encode_kernel, dim3(gx, N), dim3(256), 0, s, flow, (int64_t)sh * sw, maxd, rgb, max_out);
    LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------
// forward/backward consistency masks (bands/common/flow.py:19-40): the opposite flow is sampled at p + f(p)
// the way cv2.remap(INTER_LINEAR, BORDER_CONSTANT) does - coordinates quantised to 1/32 px with a half-even
// round, 2x2 weights as products of the float32 (1 - k/32, k/32) tables, zero outside - then
// |f + f'| < a1 (|f| + |f'|) + a2.  Every step is a separately rounded fp32 operation, as numpy evaluates it.
// flows: [n, 2, h, w, 2]; mask: [n, 2, h, w] bytes of 0 / 1.
namespace {
__device__ __forceinline__ float norm2_rn(float x, float y) {
    return ex_fsqrt(ex_fadd(ex_fmul(x, x), ex_fmul(y, y)));
}

__global__ __launch_bounds__(256) void fwdbwd_mask_kernel(const float2 *__restrict__ flow, int h, int w, float a1, float a2,
                                                          uint8_t *__restrict__ mask) {
    const int64_t hw = (int64_t)h * w;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const int nd = blockIdx.y;                                  // pair * 2 + direction
    const float2 *self = flow + (int64_t)nd * hw, *other = flow + (int64_t)(nd ^ 1) * hw;
    const int y = (int)(p / w), x = (int)(p - (int64_t)y * w);
    const float2 f = self[p];
    const int qx = __float2int_rn(ex_fmul(ex_fadd(f.x, (float)x), 32.f));
    const int qy = __float2int_rn(ex_fmul(ex_fadd(f.y, (float)y), 32.f));
    const int ix = qx >> 5, iy = qy >> 5;
    const float fx = ex_fmul((float)(qx & 31), 1.f / 32), fy = ex_fmul((float)(qy & 31), 1.f / 32);
    const float tx[2] = {ex_fsub(1.f, fx), fx}, ty[2] = {ex_fsub(1.f, fy), fy};
    float wx = 0.f, wy = 0.f;
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int yy = iy + k1, xx = ix + k2;
            float2 v = make_float2(0.f, 0.f);
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) v = other[(int64_t)yy * w + xx];
            const float wt = ex_fmul(ty[k1], tx[k2]);
            const float tx_ = ex_fmul(v.x, wt), ty_ = ex_fmul(v.y, wt);
            wx = (k1 | k2) ? ex_fadd(wx, tx_) : tx_;
            wy = (k1 | k2) ? ex_fadd(wy, ty_) : ty_;
        }
    const float err = norm2_rn(ex_fadd(f.x, wx), ex_fadd(f.y, wy));
    const float thr = ex_fadd(ex_fmul(a1, ex_fadd(norm2_rn(f.x, f.y), norm2_rn(wx, wy))), a2);
    mask[(int64_t)nd * hw + p] = err < thr ? 1 : 0;
}
}  // namespace

int launch_fwdbwd_mask(hipStream_t s, const float *flow, int n, int h, int w, float a1, float a2, uint8_t *mask) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(fwdbwd_mask_kernel, dim3(nblk((int64_t)h * w), 2 * n), dim3(256), 0, s, (const float2 *)flow, h, w, a1, a2,
                       mask);
    PB_HIP(hipGetLastError());
    return 0;
}
