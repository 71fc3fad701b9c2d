// HBM-bound kernels of the flow_gmflow band around the GEMMs and the 128-wide attention (attention128.hip).
// Reference being replaced: bands/gmflow/utils.py:61-86 (feature_add_position), transformer.py:47-101 (window split / roll / merge of
// single_head_split_window_attention), :141-181 (TransformerLayer: LayerNorm, concat, residual), matching.py:7-42 (coordinates), gmflow.py:74-92
// (upsampler input).  The token stream is fp32; every MFMA operand derived from it is a hi + lo fp16 pair (`[hi (C) | lo (C)]` per row).
#include "gmflow_kernels.h"

namespace {

inline unsigned nblk(int64_t n, int t = 256) { return (unsigned)((n + t - 1) / t); }

__device__ __forceinline__ void split8(const float (&v)[8], f16x8 &hi, f16x8 &lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { hi[j] = (f16)v[j]; lo[j] = (f16)(v[j] - (float)hi[j]); }
}

// window-order row (bw = image * 4 + wy * 2 + wx, pos = ly * ww + lx) -> row of the [images, P] token matrix; the 2 x 2 windows are cut
// from the map rolled by (-shift_y, -shift_x) (torch.roll(x, (-sy, -sx)): rolled[i] = x[(i + s) mod n])
__device__ __forceinline__ int64_t win_row(int bw, int pos, const GmGeom &g, int shifted) {
    const int img = bw >> 2, wy = (bw >> 1) & 1, wx = bw & 1;
    const int ly = pos / g.ww, lx = pos - ly * g.ww;
    int gy = wy * g.wh + ly + (shifted ? g.wh / 2 : 0), gx = wx * g.ww + lx + (shifted ? g.ww / 2 : 0);
    gy = gy >= g.h8 ? gy - g.h8 : gy;
    gx = gx >= g.w8 ? gx - g.w8 : gx;
    return (int64_t)img * g.P + gy * g.w8 + gx;
}

// X[(n, e), t] = feat[n + e, t] + pos[t]  (both frames of pair n; feature_add_position tiles ONE window's sine embedding over the 2 x 2
// windows - the table arrives tiled), and its split copy
__global__ __launch_bounds__(256) void gm_tokens_kernel(const float *__restrict__ feat, const float *__restrict__ pos, float *__restrict__ X,
                                                        f16 *__restrict__ Xs, int NP, int P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;         // (row, 8-channel chunk)
    if (i >= (int64_t)NP * 2 * P * 16) return;
    const int ch = (int)(i & 15);
    const int64_t row = i >> 4;
    const int t = (int)(row % P), img = (int)(row / P), n = img >> 1, e = img & 1;
    const float *f = feat + ((int64_t)(n + e) * P + t) * 128 + ch * 8, *p = pos + (int64_t)t * 128 + ch * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = f[j] + p[j];
    float *x = X + row * 128 + ch * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = v[j];
    f16x8 hi, lo;
    split8(v, hi, lo);
    *(f16x8 *)(Xs + row * 256 + ch * 8) = hi;
    *(f16x8 *)(Xs + row * 256 + 128 + ch * 8) = lo;
}

// fp32 rows [rows, C] (row stride ld) -> [hi (C) | lo (C)] fp16 rows
__global__ __launch_bounds__(256) void gm_split_rows_kernel(const float *__restrict__ src, int ld, int C8, f16 *__restrict__ dst, int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C8) return;
    const int ch = (int)(i % C8);
    const int64_t row = i / C8;
    float v[8];
    const float *s = src + row * ld + ch * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = s[j];
    f16x8 hi, lo;
    split8(v, hi, lo);
    *(f16x8 *)(dst + row * (C8 * 16) + ch * 8) = hi;
    *(f16x8 *)(dst + row * (C8 * 16) + C8 * 8 + ch * 8) = lo;
}

// Window pack: 128-column slices of an fp32 projection matrix [images * P, ld] -> the attention kernel's operands in window order:
// rows kind: dst [Bw, Lw, 256] = [hi | lo]; vt kind: dst [Bw, 2, 128, ldv] (hi rows, lo rows; keys along the row).  blockIdx.z = job.
__global__ __launch_bounds__(256) void gm_pack_kernel(GmPackJobs jobs, GmGeom g, int shifted) {
    __shared__ float tile[64][129];
    const GmPackJob jb = jobs.j[blockIdx.z];
    const int bw = blockIdx.y, p0 = blockIdx.x * 64, tid = threadIdx.x;
    if (!jb.is_vt) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = tid + it * 256, r = item >> 4, ch = item & 15;
            const int pos = p0 + r;
            if (pos >= g.Lw) continue;
            const float *s = jb.src + win_row(bw, pos, g, shifted) * jb.ld + jb.col + ch * 8;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = s[j];
            f16x8 hi, lo;
            split8(v, hi, lo);
            f16 *d = jb.dst + ((int64_t)bw * g.Lw + pos) * 256 + ch * 8;
            *(f16x8 *)d = hi;
            *(f16x8 *)(d + 128) = lo;
        }
        return;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {                         // 64 positions x 32 float4
        const int item = tid + it * 256, r = item >> 5, c4 = item & 31;
        const int pos = p0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pos < g.Lw) v = *(const f32x4 *)(jb.src + win_row(bw, pos, g, shifted) * jb.ld + jb.col + c4 * 4);
        tile[r][c4 * 4 + 0] = v[0]; tile[r][c4 * 4 + 1] = v[1]; tile[r][c4 * 4 + 2] = v[2]; tile[r][c4 * 4 + 3] = v[3];
    }
    __syncthreads();
    const int d = tid >> 1, half = tid & 1;                  // channel d, positions half * 32 .. + 32 (ldv is a multiple of 32: no tail test)
    f16 *vh = jb.dst + (((int64_t)bw * 2 + 0) * 128 + d) * g.ldv + p0 + half * 32;
    f16 *vl = jb.dst + (((int64_t)bw * 2 + 1) * 128 + d) * g.ldv + p0 + half * 32;
    if (p0 + half * 32 >= g.ldv) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tile[half * 32 + q * 8 + j][d];
        f16x8 hi, lo;
        split8(v, hi, lo);
        *(f16x8 *)(vh + q * 8) = hi;
        *(f16x8 *)(vl + q * 8) = lo;
    }
}

// One wave per row of M [rows, 128]: y = LayerNorm(M[row]) (eps 1e-5, biased variance, fp32 - nn.LayerNorm) and
//   mode 0: X[g] += y; Xs[g] = split(X[g])                        (TransformerLayer.forward :178-181 `source + message`)
//   mode 1: cat[g] = split([X[g] | y])   (512 halfs: hi of X, hi of y, lo of X, lo of y)   (:173 the concat that feeds the MLP)
// g = row, or with `windowed` the token row of window-order row `row`.
__global__ __launch_bounds__(256) void gm_ln_kernel(const float *__restrict__ M, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                    float *__restrict__ X, f16 *__restrict__ out, int64_t rows, GmGeom g, int windowed,
                                                    int shifted, int mode) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const f32x2 v = *(const f32x2 *)(M + row * 128 + lane * 2);
    const float mean = wave_sum(v[0] + v[1]) * (1.f / 128.f);
    const float d0 = v[0] - mean, d1 = v[1] - mean;
    const float var = wave_sum(d0 * d0 + d1 * d1) * (1.f / 128.f);
    const float rs = 1.f / sqrtf(var + 1e-5f);
    const float y0 = d0 * rs * gamma[lane * 2] + beta[lane * 2], y1 = d1 * rs * gamma[lane * 2 + 1] + beta[lane * 2 + 1];
    const int64_t gr = windowed ? win_row((int)(row / g.Lw), (int)(row % g.Lw), g, shifted) : row;
    float *x = X + gr * 128 + lane * 2;
    if (mode == 0) {
        const float n0 = x[0] + y0, n1 = x[1] + y1;
        x[0] = n0; x[1] = n1;
        const f16 h0 = (f16)n0, h1 = (f16)n1;
        f16x2 hi = {h0, h1}, lo = {(f16)(n0 - (float)h0), (f16)(n1 - (float)h1)};
        *(f16x2 *)(out + gr * 256 + lane * 2) = hi;
        *(f16x2 *)(out + gr * 256 + 128 + lane * 2) = lo;
    } else {
        const float s0 = x[0], s1 = x[1];
        const f16 a0 = (f16)s0, a1 = (f16)s1, b0 = (f16)y0, b1 = (f16)y1;
        f16 *o = out + gr * 512 + lane * 2;
        *(f16x2 *)o = f16x2{a0, a1};
        *(f16x2 *)(o + 128) = f16x2{b0, b1};
        *(f16x2 *)(o + 256) = f16x2{(f16)(s0 - (float)a0), (f16)(s1 - (float)a1)};
        *(f16x2 *)(o + 384) = f16x2{(f16)(y0 - (float)b0), (f16)(y1 - (float)b1)};
    }
}

// Vt of the global matching (matching.py:19-27): the pixel coordinates (x, y) of every key, rows 0 and 1 of a [2, 32, ldv] block
// (integers below 2048: exact in fp16, so the lo rows stay zero)
__global__ void gm_grid_vt_kernel(f16 *__restrict__ vt, int P, int w8, int ldv) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P) return;
    vt[t] = (f16)(float)(t % w8);
    vt[ldv + t] = (f16)(float)(t / w8);
}

// matching output O [B, P, 32] (expected coordinates in columns 0, 1) -> flow = correspondence - own coordinate (matching.py:29-31),
// and the propagation's Vt [B, 2, 32, ldv] = this flow as hi + lo rows
__global__ void gm_match_flow_kernel(const float *__restrict__ O, float *__restrict__ flow, f16 *__restrict__ vt, int B, int P, int w8, int ldv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * P) return;
    const int t = (int)(i % P), b = (int)(i / P);
    const float u = O[i * 32] - (float)(t % w8), v = O[i * 32 + 1] - (float)(t / w8);
    flow[i * 2] = u; flow[i * 2 + 1] = v;
    f16 *d = vt + (int64_t)b * 64 * ldv + t;
    const f16 uh = (f16)u, vh = (f16)v;
    d[0] = uh; d[ldv] = vh;
    d[32 * (int64_t)ldv] = (f16)(u - (float)uh); d[33 * (int64_t)ldv] = (f16)(v - (float)vh);
}

// propagation output O [B, P, 32] -> flow [B, P, 2] and the upsampler's input map cat(flow, feature) (gmflow.py:79) as a split NHWC map
// [B, P, 384] = [hi (192: flow 2, feature 128, zeros) | lo (192)]; the feature of batch element b is token image b * img_step
__global__ __launch_bounds__(256) void gm_upsampler_in_kernel(const float *__restrict__ O, const float *__restrict__ X, float *__restrict__ flow,
                                                              f16 *__restrict__ map, int B, int P, int img_step) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;         // (pixel, 8-channel chunk of 24)
    if (i >= (int64_t)B * P * 24) return;
    const int ch = (int)(i % 24);
    const int64_t px = i / 24;
    const int t = (int)(px % P), b = (int)(px / P);
    const float *x = X + ((int64_t)b * img_step * P + t) * 128;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = ch * 8 + j;
        v[j] = c < 2 ? O[px * 32 + c] : (c < 130 ? x[c - 2] : 0.f);
    }
    if (ch == 0) { flow[px * 2] = v[0]; flow[px * 2 + 1] = v[1]; }
    f16x8 hi, lo;
    split8(v, hi, lo);
    *(f16x8 *)(map + px * 384 + ch * 8) = hi;
    *(f16x8 *)(map + px * 384 + 192 + ch * 8) = lo;
}

}  // namespace

int launch_gm_tokens(hipStream_t s, const float *feat, const float *pos, float *X, f16 *Xs, int NP, int P) {
    hipLaunchKernelGGL(gm_tokens_kernel, dim3(nblk((int64_t)NP * 2 * P * 16)), dim3(256), 0, s, feat, pos, X, Xs, NP, P);
    PB_HIP(hipGetLastError());
    return 0;
}
int launch_gm_split_rows(hipStream_t s, const float *src, int ld, int C, f16 *dst, int64_t rows) {
    PB_CHECK(C % 8 == 0, PB_ERR_ARG, "gm_split_rows: C = %d", C);
    hipLaunchKernelGGL(gm_split_rows_kernel, dim3(nblk(rows * (C / 8))), dim3(256), 0, s, src, ld, C / 8, dst, rows);
    PB_HIP(hipGetLastError());
    return 0;
}
int launch_gm_pack(hipStream_t s, const GmPackJobs &jobs, const GmGeom &g, int Bw, int shifted) {
    PB_CHECK(jobs.n >= 1 && jobs.n <= 5, PB_ERR_ARG, "gm_pack: %d jobs", jobs.n);
    hipLaunchKernelGGL(gm_pack_kernel, dim3((g.ldv + 63) / 64, Bw, jobs.n), dim3(256), 0, s, jobs, g, shifted);
    PB_HIP(hipGetLastError());
    return 0;
}
int launch_gm_ln(hipStream_t s, const float *M, const float *gamma, const float *beta, float *X, f16 *out, int64_t rows, const GmGeom &g,
                 int windowed, int shifted, int mode) {
    hipLaunchKernelGGL(gm_ln_kernel, dim3(nblk(rows, 4)), dim3(256), 0, s, M, gamma, beta, X, out, rows, g, windowed, shifted, mode);
    PB_HIP(hipGetLastError());
    return 0;
}
int launch_gm_grid_vt(hipStream_t s, f16 *vt, int P, int w8, int ldv) {
    hipLaunchKernelGGL(gm_grid_vt_kernel, dim3(nblk(P)), dim3(256), 0, s, vt, P, w8, ldv);
    PB_HIP(hipGetLastError());
    return 0;
}
int launch_gm_match_flow(hipStream_t s, const float *O, float *flow, f16 *vt, int B, int P, int w8, int ldv) {
    hipLaunchKernelGGL(gm_match_flow_kernel, dim3(nblk((int64_t)B * P)), dim3(256), 0, s, O, flow, vt, B, P, w8, ldv);
    PB_HIP(hipGetLastError());
    return 0;
}
int launch_gm_upsampler_in(hipStream_t s, const float *O, const float *X, float *flow, f16 *map, int B, int P, int img_step) {
    hipLaunchKernelGGL(gm_upsampler_in_kernel, dim3(nblk((int64_t)B * P * 24)), dim3(256), 0, s, O, X, flow, map, B, P, img_step);
    PB_HIP(hipGetLastError());
    return 0;
}
