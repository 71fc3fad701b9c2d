// fp16 MFMA GEMM and implicit-GEMM convolution kernels for gfx950 (CDNA4).
//
// One kernel template serves every matmul-shaped op of the bands engine: the ViT linears
// (qkv / proj / fc1 / fc2, reference dinov2/layers/attention.py:49-62, mlp.py:35-41), the patch
// embedding (patch_embed.py:66-82), and the DPT head's 1x1 / 3x3 / transposed convolutions
// (bands/d_anything/dpt.py:103-136, blocks.py:69-153) as NHWC implicit GEMM.
//
// Structure (per workgroup): BM x BN output tile, K in steps of 64 halfs.
//   * A and W tiles go HBM -> LDS with global_load_lds_dwordx4 (16 B per lane, no VGPR round
//     trip), double buffered, one barrier per K step; the next step's loads are issued before
//     the current step's MFMAs.
//   * LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7, applied
//     on the global SOURCE address (the LDS image of a DMA is lane-linear) and again on the
//     ds_read_b128 fragment reads, which makes those reads bank-conflict free.
//   * v_mfma_f32_32x32x16_f16, fp32 accumulators; wave tile (BM/WM) x (BN/WN).
//   * Epilogue: accumulators -> per-wave LDS patch -> each lane owns 8 consecutive columns of
//     one row (16-byte stores), with the op-specific fusion (bias, GELU, ReLU, LayerScale +
//     residual, q/k/v split with V transposed, pixel-shuffle for transposed convs, ...).
//   * blockIdx is remapped so that each XCD (private L2) works on a contiguous range of tiles.
#pragma once
#include "gemm.h"

#include <stdlib.h>

#include <type_traits>

namespace {

// All CUs start their first tile together and every tile of a launch takes the same time, so without help every
// CU reaches its epilogue at the same moment: the whole chip writes 256 x (128..512 KB) at once and each epilogue
// lasts as long as that HBM burst (measured 14k-38k cycles against a 51k-cycle main loop), while HBM idles during
// the main loops.  Delaying the first-wave workgroups by eighths of a tile period de-phases the CUs for the rest
// of the launch; stores then drain under other CUs' MFMA time.
__device__ __forceinline__ void stagger_start(int stagger, int first_wave) {
    if (stagger > 0 && (int)blockIdx.x < first_wave) {
        const int n = ((blockIdx.x >> 3) & 7) * stagger;     // units of 64 cycles
        for (int i = 0; i < n; i += 100) __builtin_amdgcn_s_sleep(100);
    }
}

// ---- fused epilogues ---------------------------------------------------------------------------
// A lane owns the same 8 output columns (n .. n+7) for every row it stores, so everything that
// depends only on the column (bias, LayerScale gamma, head weights) is loaded ONCE per lane.  Row
// dependent operands (residual stream, skip tensors, pos-embed) are fetched for all of a pass's
// rows before the first store: the compiler cannot hoist a load above a store that may alias it, and
// a load -> use -> store chain per row costs one full memory latency each (measured 23k-53k
// cycles per 256x256 tile before this restructuring, 2-3x the MFMA main loop's share).
typedef int i32x8 __attribute__((ext_vector_type(8)));
// two 16-byte fragments (the ones the fp16 path feeds to k-steps 2q and 2q + 1) as the 32-byte operand of one fp8 MFMA: any
// assignment of the tile's bytes to (lane half, byte slot) is valid as long as A and B use the same one (tools/probe/mx_probe.hip)
__device__ __forceinline__ i32x8 cat_frag(f16x8 a, f16x8 b) {
    union { f16x8 h[2]; i32x8 v; } u;
    u.h[0] = a; u.h[1] = b;
    return u.v;
}
__device__ __forceinline__ f32x16 mfma_mx8(f16x8 a0, f16x8 a1, f16x8 b0, f16x8 b1, f32x16 c, int sa, int sb) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat_frag(a0, a1), cat_frag(b0, b1), c, 0, 0, 0, sa, 0, sb);
}

// split maps with e4m3 residual parts (gemm.h lo8): per pixel [hi fp16 (C) | hi8 (C bytes) | lo8 (C bytes)], C = lo_off.
// `pix` = element offset of the pixel's first half, `col` = channel.
__device__ __forceinline__ void lo8_store2(f16 *base, int64_t pix, int col, int C, float v0, float v1, float s_hi, float s_lo) {
    const f16 h0 = (f16)v0, h1 = (f16)v1;
    char *b = (char *)(base + pix);
    *(unsigned short *)(b + 2 * C + col) = pb_fp8x2((float)h0 * s_hi, (float)h1 * s_hi);
    *(unsigned short *)(b + 3 * C + col) = pb_fp8x2((v0 - (float)h0) * s_lo, (v1 - (float)h1) * s_lo);
}
__device__ __forceinline__ f32x2 lo8_load2(const f16 *base, int64_t pix, int col, int C, float inv_lo) {
    const unsigned short u = *(const unsigned short *)((const char *)(base + pix) + 3 * C + col);
    const f32x2 f = __builtin_amdgcn_cvt_pk_f32_fp8((int)u, false);
    return f * inv_lo;
}

struct EpiAux {
    f16x8 a1, a2;       // EPI_STD skip tensors
    f32x4 r0, r1;       // EPI_RESID residual / EPI_PATCH pos-embed
};

__device__ __forceinline__ float fast_gelu(float x) {
    // exact-erf GELU as  max(x, 0) - |x| * Q(|x|),  Q(a) = 0.5 erfc(a / sqrt 2) = 2^P(a): a degree-5 minimax fit of
    // log2 Q on [0, 6.5] weighted by a Q(a) (tools: numpy lstsq, max |error| of the whole expression 6.4e-7 in fp32
    // Horner arithmetic, i.e. below fp32 round-off of the result for |x| > 4).  5 FMAs + one v_exp_f32; the A&S
    // 7.1.26 form needed an rcp and an exp and twice the VALU work (14k of a 75k-cycle fc1 tile).
    const float a = fminf(fabsf(x), 6.5f);
    float p = fmaf(a, -0.00047330817324109375f, 0.007084541954100132f);
    p = fmaf(a, p, -0.051827322691679f);
    p = fmaf(a, p, -0.45999252796173096f);
    p = fmaf(a, p, -1.1507878303527832f);
    p = fmaf(a, p, -1.000037670135498f);
    return fmaxf(x, 0.f) - a * __builtin_amdgcn_exp2f(p);
}

// tanh as 1 - 2 / (1 + e^(2x)): one v_exp_f32, one v_rcp_f32 and three plain operations, saturating correctly at both ends (e^(2x) -> inf
// gives 1, -> 0 gives -1).  The device library's tanhf is ~45 instructions with a divergent branch per element; the GRU's q gate evaluates
// 73 M of them per launch.  Absolute error <= 2e-7 (cancellation near 0 costs RELATIVE accuracy only where tanh itself is ~0): far below the
// fp16 rounding of the state that consumes it.
__device__ __forceinline__ float fast_tanh(float x) {
    return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x));
}

// two values per instruction where the ISA has a packed form (v_pk_fma_f32): the fc1 epilogue is VALU bound
__device__ __forceinline__ void fast_gelu2(float &x0, float &x1) {
    // v_med3_f32 clamps without the canonicalising v_max that fminf / fmaxf put in front of every operand
    const f32x2 a = {__builtin_amdgcn_fmed3f(fabsf(x0), 0.f, 6.5f), __builtin_amdgcn_fmed3f(fabsf(x1), 0.f, 6.5f)};
    const f32x2 c5 = {-0.00047330817324109375f, -0.00047330817324109375f}, c4 = {0.007084541954100132f, 0.007084541954100132f},
                c3 = {-0.051827322691679f, -0.051827322691679f}, c2 = {-0.45999252796173096f, -0.45999252796173096f},
                c1 = {-1.1507878303527832f, -1.1507878303527832f}, c0 = {-1.000037670135498f, -1.000037670135498f};
    f32x2 p = __builtin_elementwise_fma(a, c5, c4);
    p = __builtin_elementwise_fma(a, p, c3);
    p = __builtin_elementwise_fma(a, p, c2);
    p = __builtin_elementwise_fma(a, p, c1);
    p = __builtin_elementwise_fma(a, p, c0);
    const f32x2 e = {__builtin_amdgcn_exp2f(p[0]), __builtin_amdgcn_exp2f(p[1])};
    const f32x2 r = {__builtin_amdgcn_fmed3f(x0, 0.f, 3.0e38f), __builtin_amdgcn_fmed3f(x1, 0.f, 3.0e38f)};
    const f32x2 o = __builtin_elementwise_fma(-a, e, r);
    x0 = o[0]; x1 = o[1];
}

template <int EPI>
__device__ __forceinline__ void epi_cols(const GemmArgs &p, int n, bool nok, float (&cb)[8], float (&cg)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { cb[j] = 0.f; cg[j] = 0.f; }
    if (!nok) return;
    if constexpr (EPI == EPI_PIXSHUF) {
        const int co = n % p.ps_co;
#pragma unroll
        for (int j = 0; j < 8; ++j) cb[j] = p.bias[co + j];
    } else {
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < 8; ++j) cb[j] = p.bias[n + j];
        }
    }
    if constexpr (EPI == EPI_RESID) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cg[j] = p.gamma[n + j];
    }
    if constexpr (EPI == EPI_HEAD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cg[j] = p.w2[n + j];
    }
}

template <int EPI>
__device__ __forceinline__ void epi_prefetch(const GemmArgs &p, int m, int n, EpiAux &x) {
    if constexpr (EPI == EPI_STD) {
        const int64_t o = (int64_t)m * p.ldo + n;
        if (p.add1) x.a1 = *(const f16x8 *)(p.add1 + o);
        if (p.add2) x.a2 = *(const f16x8 *)(p.add2 + o);
    } else if constexpr (EPI == EPI_RESID) {
        const float *r = p.resid + (int64_t)m * p.ldr + n;
        x.r0 = *(const f32x4 *)r;
        x.r1 = *(const f32x4 *)(r + 4);
    } else if constexpr (EPI == EPI_PATCH) {
        const int b = m / p.ppi, pi = m - b * p.ppi;
        const float *pe = p.pos + (int64_t)(1 + pi) * p.D + n;
        x.r0 = *(const f32x4 *)pe;
        x.r1 = *(const f32x4 *)(pe + 4);
    }
}

template <int EPI>
__device__ __forceinline__ void epi_finish(const GemmArgs &p, int m, int n, float (&v)[8], const EpiAux &x,
                                           const float (&cb)[8], const float (&cg)[8]) {
    if constexpr (EPI == EPI_STD) {
        const int64_t o = (int64_t)m * p.ldo + n;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += cb[j];
        if (p.pre_relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (p.add1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)x.a1[j];
            if (p.lo_off) {
                const f16x8 l = *(const f16x8 *)(p.add1 + o + p.lo_off);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)l[j];
            }
        }
        if (p.add2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)x.a2[j];
            if (p.lo_off) {
                const f16x8 l = *(const f16x8 *)(p.add2 + o + p.lo_off);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)l[j];
            }
        }
        if (p.out) {
            f16x8 r;
            if (p.act == ACT_GELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = (f16)fast_gelu(v[j]);
            } else if (p.act == ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = (f16)fmaxf(v[j], 0.f);
            } else if (p.act == ACT_SIGMOID) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = (f16)(1.f / (1.f + __expf(-v[j])));
            } else if (p.act == ACT_TANH) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = (f16)fast_tanh(v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = (f16)v[j];
            }
            *(f16x8 *)(p.out + o) = r;
            if (p.lo_off) {                  // (only the linear / ReLU activations are used with split outputs)
                f16x8 l;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = p.act == ACT_RELU ? fmaxf(v[j], 0.f) : v[j];
                    l[j] = (f16)(t - (float)(f16)t);
                }
                *(f16x8 *)(p.out + o + p.lo_off) = l;
            }
        }
        if (p.out2) {
            f16x8 r;
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = (f16)fmaxf(v[j], 0.f);
            *(f16x8 *)(p.out2 + o) = r;
            if (p.lo_off) {
                f16x8 l;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float t = fmaxf(v[j], 0.f); l[j] = (f16)(t - (float)(f16)t); }
                *(f16x8 *)(p.out2 + o + p.lo_off) = l;
            }
        }
    } else if constexpr (EPI == EPI_F32) {
        float *r = p.out32 + (int64_t)m * p.ldo + n;
        f32x4 r0, r1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r0[j] = (v[j] + cb[j]) * p.scale;
            r1[j] = (v[4 + j] + cb[4 + j]) * p.scale;
        }
        *(f32x4 *)r = r0;
        *(f32x4 *)(r + 4) = r1;
    } else if constexpr (EPI == EPI_RESID) {
        float *r = p.resid + (int64_t)m * p.ldr + n;
        f32x4 r0 = x.r0, r1 = x.r1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r0[j] += cg[j] * (v[j] + cb[j]);
            r1[j] += cg[4 + j] * (v[4 + j] + cb[4 + j]);
        }
        *(f32x4 *)r = r0;
        *(f32x4 *)(r + 4) = r1;
    } else if constexpr (EPI == EPI_QKV) {
        // q / k rows only; the V third is handled by the transposed path below
        const int which = n / p.D;
        const int hn = n - which * p.D;
        const int head = hn >> 6, d = hn & 63;
        const int b = m / p.ntp, t = m - b * p.ntp;
        const float s = which == 0 ? p.qscale : 1.f;
        f16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (f16)((v[j] + cb[j]) * s);
        f16 *dst = (which == 0 ? p.q : p.k) + (((int64_t)b * p.heads + head) * p.ntp + t) * 64 + d;
        *(f16x8 *)dst = r;
    } else if constexpr (EPI == EPI_PIXSHUF) {
        const int hw = p.ps_h * p.ps_w;
        const int b = m / hw, rem = m - b * hw;
        const int y = rem / p.ps_w, xx = rem - y * p.ps_w;
        const int tap = n / p.ps_co, co = n - tap * p.ps_co;
        const int dy = tap / p.ps_s, dx = tap - dy * p.ps_s;
        const int64_t row = ((int64_t)b * p.ps_h * p.ps_s + (y * p.ps_s + dy)) * (p.ps_w * p.ps_s) + (xx * p.ps_s + dx);
        f16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (f16)(v[j] + cb[j]);
        *(f16x8 *)(p.out + row * p.ldo + co) = r;
        if (p.lo_off && p.lo8) {
            const float shi = __builtin_ldexpf(1.f, p.lo8_pa), slo = __builtin_ldexpf(1.f, p.lo8_pa + 12);
#pragma unroll
            for (int j = 0; j < 8; j += 2) lo8_store2(p.out, row * p.ldo, co + j, p.lo_off, v[j] + cb[j], v[j + 1] + cb[j + 1], shi, slo);
        } else if (p.lo_off) {
            f16x8 l;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float t = v[j] + cb[j]; l[j] = (f16)(t - (float)(f16)t); }
            *(f16x8 *)(p.out + row * p.ldo + co + p.lo_off) = l;
        }
    } else if constexpr (EPI == EPI_PATCH) {
        const int b = m / p.ppi, pi = m - b * p.ppi;
        float *r = p.resid + ((int64_t)b * p.ntp + 1 + pi) * p.ldr + n;
        f32x4 r0, r1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r0[j] = v[j] + cb[j] + x.r0[j];
            r1[j] = v[4 + j] + cb[4 + j] + x.r1[j];
        }
        *(f32x4 *)r = r0;
        *(f32x4 *)(r + 4) = r1;
    }
}

// EPI_RESID: X += A W'^T + b' with LayerScale pre-folded into W' and b'.  The accumulators START from the fp32
// residual tile (loaded in the prologue, under the first DMAs' latency) and are stored straight back from the
// MFMA layout: no LDS transpose, no read in the epilogue.  Round 4: the weight rows are fed in interleaved column order
// (epi_interleaved), so a lane owns columns 2 li, 2 li + 1 of every row and moves them as ONE 8-byte access - a half wave
// covers 256 contiguous bytes and a 128 x 64 wave tile takes 64 instructions instead of 128 - through a buffer resource over
// the residual stream: the row term of the address is a scalar (soffset), the lane term one loop-invariant VGPR, where the
// flat form spent a 64-bit multiply-add per access and kept 128 address pairs alive (profiles/r04a_store_probe.txt prices the patterns).
// Rows >= M / columns >= N get an out-of-range offset: such loads return 0 and such stores are dropped.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int TM, bool STORE, bool CHECK, bool ATOMIC = false>
__device__ __forceinline__ void resid_io_impl(const GemmArgs &p, f32x16 (&acc)[TM][2], int wave_m0, int wave_n0, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const int n = wave_n0 + 2 * li;
    const bool nok = !CHECK || n < p.N;
    const int nc = nok ? n : 0;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.resid, (unsigned)((int64_t)p.M * p.ldr * 4));
    const int ldr = (int)p.ldr;
    const unsigned voff = (unsigned)(((wave_m0 + 4 * lh) * ldr + nc) * 4);
    float b0 = 0.f, b1 = 0.f;
    if (!STORE) { b0 = p.bias[nc]; b1 = p.bias[nc + 1]; }
    if constexpr (ATOMIC && !STORE) {                    // X += (b' + A W'^T): the accumulators start from the bias, nothing is loaded
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[tm][0][r] = b0; acc[tm][1][r] = b1; }
        return;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rowc = tm * 32 + (r & 3) + 8 * (r >> 2);
            const int soff = rowc * ldr * 4;
            unsigned vo = voff;
            if (CHECK) vo = (nok && wave_m0 + rowc + 4 * lh < p.M) ? voff : 0xFFFFFF00u;
            if constexpr (ATOMIC) {
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[tm][0][r], rs, (int)vo, soff, 0);
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[tm][1][r], rs, (int)vo + 4, soff, 0);
            } else if (STORE) {
                const f32x2 o = {acc[tm][0][r], acc[tm][1][r]};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rs, (int)vo, soff, 0);
            } else {
                const f32x2 x = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)vo, soff, 0));
                acc[tm][0][r] = x[0] + b0;
                acc[tm][1][r] = x[1] + b1;
            }
        }
}
template <int TM, int TN, bool STORE>
__device__ __forceinline__ void resid_io(const GemmArgs &p, f32x16 (&acc)[TM][TN], int wave_m0, int wave_n0, int lane) {
    static_assert(TN == 2, "the residual epilogue is built for two 32-column MFMA tiles per wave");
    if (wave_m0 + TM * 32 <= p.M && wave_n0 + 64 <= p.N) resid_io_impl<TM, STORE, false>(p, acc, wave_m0, wave_n0, lane);
    else resid_io_impl<TM, STORE, true>(p, acc, wave_m0, wave_n0, lane);
}
// persistent workgroups: store this tile's residual rows and request the NEXT tile's into the same registers, access by access - the
// write burst and the read burst of a tile boundary overlap in the memory system instead of following each other (the loads of a
// separate pass queue behind every store: vmcnt is one in-order counter).  Interior tiles only (no out-of-range handling).
template <int TM>
__device__ __forceinline__ void resid_swap(const GemmArgs &p, f32x16 (&acc)[TM][2], int cur_m0, int cur_n0, int nxt_m0, int nxt_n0, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.resid, (unsigned)((int64_t)p.M * p.ldr * 4));
    const int ldr = (int)p.ldr;
    const unsigned vc = (unsigned)(((cur_m0 + 4 * lh) * ldr + cur_n0 + 2 * li) * 4), vn = (unsigned)(((nxt_m0 + 4 * lh) * ldr + nxt_n0 + 2 * li) * 4);
    const float b0 = p.bias[nxt_n0 + 2 * li], b1 = p.bias[nxt_n0 + 2 * li + 1];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int soff = (tm * 32 + (r & 3) + 8 * (r >> 2)) * ldr * 4;
            const f32x2 o = {acc[tm][0][r], acc[tm][1][r]};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rs, (int)vc, soff, 0);
            const f32x2 x = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)vn, soff, 0));
            acc[tm][0][r] = x[0];
            acc[tm][1][r] = x[1];
        }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[tm][0][r] += b0; acc[tm][1][r] += b1; }
}
template <int TM, int TN, bool STORE>
__device__ __forceinline__ void resid_io_atomic(const GemmArgs &p, f32x16 (&acc)[TM][TN], int wave_m0, int wave_n0, int lane) {
    if (wave_m0 + TM * 32 <= p.M && wave_n0 + 64 <= p.N) resid_io_impl<TM, STORE, false, true>(p, acc, wave_m0, wave_n0, lane);
    else resid_io_impl<TM, STORE, true, true>(p, acc, wave_m0, wave_n0, lane);
}

// ---- interleaved output columns (fp16 epilogues, TN == 2) -----------------------------------------
// A global store instruction is cheapest when each half wave writes ONE contiguous 128-byte line (measured:
// ~11 cycles per instruction and CU for 2 x 128 B against ~140 for the 8 x 128 B pattern of the LDS-transposed
// epilogue, which made a 256 x 256 fp16 tile cost 18-22k cycles).  In the MFMA layout a lane owns column `li`
// of each 32-wide tile, so the two tiles of a wave are fed weight rows in the order  LDS row (tn*32 + j) <-
// column 2j + tn : lane li then holds columns 2 li and 2 li + 1 of a row in acc[.][0] / acc[.][1], packs them
// into one dword, and 32 lanes cover 64 consecutive fp16 columns = 128 bytes.  Only the DMA source rows of
// the weight operand change; nothing moves between lanes.
// TN == 3 (the 128 x 96 tile, EPI_STD): the first two column blocks of a wave are such a pair, the third keeps its natural order (col_map3)
template <int EPI, int TN>
__host__ __device__ constexpr bool epi_interleaved() {
    return (TN == 2 && (EPI == EPI_STD || EPI == EPI_QKV || EPI == EPI_PIXSHUF || EPI == EPI_RESID)) || (TN == 3 && EPI == EPI_STD);
}
#include "pixshuf_walk.h"   // pixshuf_first_pixel / pixshuf_wraps: plain C++, also compiled on the host by tests/test_pixshuf_walk_cpu.py
#include "conv_walk.h"      // tap_range / tap_mask / conv_ktab_word / ktab_bytes / ktab_sel: plain C++, also compiled on the host by tests/test_conv_walk_cpu.py
constexpr int KTAB_BYTES = 2048;                                          // K <= 512 tiles (checked by the launchers)
constexpr int KTAB_CW_BYTES = 4096;                                       // the 128 x 96 tile's chunk table: 8 words per tile, K <= 128 tiles
__device__ __forceinline__ unsigned conv_ktab_entry(const GemmArgs &p, int cld, int t) {
    return conv_ktab_word(p.cTapInner, p.cKH, p.cKW, p.cC, p.cW, cld, p.kwrap, p.kshift, t);
}

__device__ __forceinline__ int col_map(int r, bool il) {        // tile-local B row -> tile-local output column
    return il ? (r & ~63) + 2 * (r & 31) + ((r >> 5) & 1) : r;
}
template <int TN>
__device__ __forceinline__ int col_map_tn(int r, bool il) {     // ... of a wave that owns TN column blocks: an odd third block is not part of a pair
    return (TN == 3 && r >= 64) ? r : col_map(r, il);
}

// The third column block of a 128 x 96 tile's wave (TN == 3): a lane owns ONE column (li) and 16 rows of it.  EPI_STD with what the layers that
// have 64 < N <= 96 use (RAFT / GMFlow encoder stage 2: bias, ReLU in front of / behind one or two skip tensors, plain / [hi | lo] / [hi | hi8 | lo8]
// maps) - launch_gemm refuses the tile for anything else.  Same arithmetic, in the same order, as direct_epilogue_*: a launch's bytes do not
// depend on the tile it ran on (tests/test_gpu_ops.py).
template <int TM, bool MX>
__device__ __forceinline__ void single_col_epilogue(const GemmArgs &p, f32x16 (&acc)[TM][3], int wave_m0, int col0, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const int n = col0 + li;
    if (n >= p.N) return;
    const float b = p.bias ? p.bias[n] : 0.f;
    const int C = p.lo_off;
    const float s_hi = __builtin_ldexpf(1.f, p.lo8_pa), s_lo = __builtin_ldexpf(1.f, p.lo8_pa + 12), inv_lo = __builtin_ldexpf(1.f, -(p.lo8_pa + 12));
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        int64_t pix[16];
        float v[16];
        bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wave_m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            ok[r] = m < p.M;
            pix[r] = (int64_t)(ok[r] ? m : p.M - 1) * p.ldo;
            v[r] = acc[tm][2][r] + b;
        }
        if (p.pre_relu) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const f16 *add = k == 0 ? p.add1 : p.add2;
            if (!add) continue;
            f16 h[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) h[r] = add[pix[r] + n];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += (float)h[r];
            if (C) {
                if constexpr (MX) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned char u = *((const unsigned char *)(add + pix[r]) + 3 * C + n);
                        v[r] += __builtin_amdgcn_cvt_f32_fp8((int)u, 0) * inv_lo;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) h[r] = add[pix[r] + n + C];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += (float)h[r];
                }
            }
        }
        if (p.act == ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (!ok[r]) continue;
            const f16 h = (f16)v[r];
            p.out[pix[r] + n] = h;
            if (C) {
                if constexpr (MX) {
                    unsigned char *bp = (unsigned char *)(p.out + pix[r]);
                    bp[2 * C + n] = (unsigned char)pb_fp8x2((float)h * s_hi, 0.f);
                    bp[3 * C + n] = (unsigned char)pb_fp8x2((v[r] - (float)h) * s_lo, 0.f);
                } else {
                    p.out[pix[r] + n + C] = (f16)(v[r] - (float)h);
                }
            }
        }
    }
}

// LOM: 0 plain outputs, 1 split maps with fp16 residuals ([hi | lo]), 2 split maps with e4m3 residual parts (gemm.h lo8); each mode
// is its own copy of the code so that the plain path keeps its register allocation
template <int EPI, int TM, bool CHECK, int LOM>
__device__ __forceinline__ void direct_epilogue_f16_impl(const GemmArgs &p, f32x16 (&acc)[TM][2], int wave_m0, int wave_n0, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const int n = wave_n0 + 2 * li;
    const bool nok = n < p.N;
    const int nc = nok ? n : 0;
    float b0 = 0.f, b1 = 0.f;
    if (p.bias) {
        const int bi = EPI == EPI_PIXSHUF ? nc % p.ps_co : nc;
        b0 = p.bias[bi]; b1 = p.bias[bi + 1];
    }
    // per-wave constants of the split / shuffle epilogues (a wave's 64 columns never straddle a head or a tap)
    f16 *qk_base = nullptr;
    float qs = 1.f;
    int tap_dy = 0, tap_dx = 0, co = 0;
    if constexpr (EPI == EPI_QKV) {
        const int which = wave_n0 / p.D, hn = wave_n0 - which * p.D;
        qk_base = (which == 0 ? p.q : p.k) + (int64_t)(hn >> 6) * p.ntp * 64 + 2 * li;
        qs = which == 0 ? p.qscale : 1.f;
    }
    if constexpr (EPI == EPI_PIXSHUF) {
        const int tap = wave_n0 / p.ps_co;
        co = wave_n0 - tap * p.ps_co + 2 * li;
        tap_dy = tap / p.ps_s; tap_dx = tap - tap_dy * p.ps_s;
    }
    // lo8 (LO only): this lane's channel inside the pixel and the power-of-two scalings of the e4m3 parts
    const int pcol = EPI == EPI_PIXSHUF ? co : nc;
    const float lo8_shi = __builtin_ldexpf(1.f, p.lo8_pa), lo8_slo = __builtin_ldexpf(1.f, p.lo8_pa + 12), lo8_inv = __builtin_ldexpf(1.f, -(p.lo8_pa + 12));
#pragma unroll
    for (int th = 0; th < TM * 2; ++th) {                   // 8 accumulator registers (= 16 rows) per pass
        const int tm = th >> 1, r0 = (th & 1) * 8;
        int64_t off[8];
        int mr[8];
        bool ok[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = r0 + q;
            const int m = wave_m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            ok[q] = !CHECK || (nok && m < p.M);
            const int mc = (!CHECK || m < p.M) ? m : p.M - 1;
            mr[q] = mc;
            if constexpr (EPI == EPI_QKV) {
                const int b = mc / p.ntp, t = mc - b * p.ntp;
                off[q] = ((int64_t)b * p.heads * p.ntp + t) * 64;
            } else if constexpr (EPI == EPI_PIXSHUF) {
                const int hw = p.ps_h * p.ps_w;
                const int b = mc / hw, rem = mc - b * hw;
                const int y = rem / p.ps_w, x = rem - y * p.ps_w;
                off[q] = (((int64_t)b * p.ps_h * p.ps_s + (y * p.ps_s + tap_dy)) * (p.ps_w * p.ps_s) + (x * p.ps_s + tap_dx)) * p.ldo + co;
            } else {
                off[q] = (int64_t)mc * p.ldo + nc;
            }
        }
        // every runtime switch (skip tensors, ReLU'd copy, activation) wraps a whole 8-register loop, never a
        // single element: per-element branches on kernel arguments explode into hundreds of exec-mask branches
        float v0[8], v1[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { v0[q] = acc[tm][0][r0 + q] + b0; v1[q] = acc[tm][1][r0 + q] + b1; }
        if constexpr (EPI == EPI_STD) {
            if (p.pre_relu) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
            }
            if (p.add1) {
                f16x2 a[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) a[q] = *(const f16x2 *)(p.add1 + off[q]);
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] += (float)a[q][0]; v1[q] += (float)a[q][1]; }
                if constexpr (LOM == 1 || LOM == 2) {
                    if constexpr (LOM == 2) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) { const f32x2 l = lo8_load2(p.add1, off[q] - pcol, pcol, p.lo_off, lo8_inv); v0[q] += l[0]; v1[q] += l[1]; }
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) a[q] = *(const f16x2 *)(p.add1 + off[q] + p.lo_off);
#pragma unroll
                        for (int q = 0; q < 8; ++q) { v0[q] += (float)a[q][0]; v1[q] += (float)a[q][1]; }
                    }
                }
            }
            if (p.add2) {
                f16x2 a[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) a[q] = *(const f16x2 *)(p.add2 + off[q]);
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] += (float)a[q][0]; v1[q] += (float)a[q][1]; }
                if constexpr (LOM == 1 || LOM == 2) {
                    if constexpr (LOM == 2) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) { const f32x2 l = lo8_load2(p.add2, off[q] - pcol, pcol, p.lo_off, lo8_inv); v0[q] += l[0]; v1[q] += l[1]; }
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) a[q] = *(const f16x2 *)(p.add2 + off[q] + p.lo_off);
#pragma unroll
                        for (int q = 0; q < 8; ++q) { v0[q] += (float)a[q][0]; v1[q] += (float)a[q][1]; }
                    }
                }
            }
            if (p.out2) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    f16x2 o2;
                    o2[0] = (f16)fmaxf(v0[q], 0.f); o2[1] = (f16)fmaxf(v1[q], 0.f);
                    if (!CHECK || ok[q]) *(f16x2 *)(p.out2 + off[q]) = o2;
                    if constexpr (LOM == 3) {
                        if (!CHECK || ok[q])
                            *(unsigned short *)((char *)p.out2 + (off[q] - nc) * 2 + p.o8_off + nc) = pb_fp8x2((float)o2[0] * p.o8_scale, (float)o2[1] * p.o8_scale);
                    }
                }
                if constexpr (LOM == 1 || LOM == 2) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float r0 = fmaxf(v0[q], 0.f), r1 = fmaxf(v1[q], 0.f);
                        if constexpr (LOM == 2) {
                            if (!CHECK || ok[q]) lo8_store2(p.out2, off[q] - pcol, pcol, p.lo_off, r0, r1, lo8_shi, lo8_slo);
                        } else {
                            f16x2 o2;
                            o2[0] = (f16)(r0 - (float)(f16)r0); o2[1] = (f16)(r1 - (float)(f16)r1);
                            if (!CHECK || ok[q]) *(f16x2 *)(p.out2 + off[q] + p.lo_off) = o2;
                        }
                    }
                }
            }
            if (p.act == ACT_GELU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) fast_gelu2(v0[q], v1[q]);
            } else if (p.act == ACT_RELU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
            } else if (p.act == ACT_SIGMOID) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] = __frcp_rn(1.f + __expf(-v0[q])); v1[q] = __frcp_rn(1.f + __expf(-v1[q])); }
            } else if (p.act == ACT_TANH) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] = fast_tanh(v0[q]); v1[q] = fast_tanh(v1[q]); }
            } else if (p.act == ACT_GRU_ZR) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] = __frcp_rn(1.f + __expf(-v0[q])); v1[q] = __frcp_rn(1.f + __expf(-v1[q])); }
                if (wave_n0 >= 128) {            // r columns (a wave's 64 columns are all z or all r): r * h -> gru_rh
                    f32x2 h[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) h[q] = *(const f32x2 *)(p.gru_h + (int64_t)mr[q] * 128 + (nc - 128));
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        f16x2 o;
                        o[0] = (f16)(v0[q] * h[q][0]); o[1] = (f16)(v1[q] * h[q][1]);
                        if (!CHECK || ok[q]) *(f16x2 *)(p.gru_rh + (int64_t)mr[q] * p.gru_ld + (nc - 128)) = o;
                        if constexpr (LOM == 3) {       // fp8 twin of r * h for the q convolution's MX segment (o8_off = byte offset of the copy)
                            if (!CHECK || ok[q])
                                *(unsigned short *)((char *)(p.gru_rh + (int64_t)mr[q] * p.gru_ld) + p.o8_off + (nc - 128)) =
                                    pb_fp8x2((float)o[0] * p.o8_scale, (float)o[1] * p.o8_scale);
                        }
                    }
                    continue;
                }
            } else if (p.act == ACT_GRU_Q) {
                f32x2 h[8];
                f16x2 z[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    h[q] = *(const f32x2 *)(p.gru_h + (int64_t)mr[q] * 128 + nc);
                    z[q] = *(const f16x2 *)(p.gru_z + (int64_t)mr[q] * 256 + nc);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float z0 = (float)z[q][0], z1 = (float)z[q][1];
                    f32x2 hn;
                    hn[0] = (1.f - z0) * h[q][0] + z0 * fast_tanh(v0[q]);
                    hn[1] = (1.f - z1) * h[q][1] + z1 * fast_tanh(v1[q]);
                    if (!CHECK || ok[q]) *(f32x2 *)(p.gru_h + (int64_t)mr[q] * 128 + nc) = hn;
                    v0[q] = hn[0]; v1[q] = hn[1];
                }
            }
        }
        if constexpr (EPI == EPI_PIXSHUF) {
            if (p.act == ACT_RELU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
            }
        }
        f16 *dst = EPI == EPI_QKV ? qk_base : p.out;
        if (EPI != EPI_STD || p.out) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f16x2 o;
                o[0] = (f16)(v0[q] * qs); o[1] = (f16)(v1[q] * qs);
                if (!CHECK || ok[q]) *(f16x2 *)(dst + off[q]) = o;
            }
            if constexpr (EPI == EPI_STD && LOM == 3) {
                if (p.act != ACT_GRU_ZR) {       // fp8 copy for a consumer's MX correction segment (gemm.h o8_off); z itself has none
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const unsigned short o8 = pb_fp8x2(v0[q] * p.o8_scale, v1[q] * p.o8_scale);
                        if (!CHECK || ok[q]) *(unsigned short *)((char *)p.out + (off[q] - nc) * 2 + p.o8_off + nc) = o8;
                    }
                }
            }
            if constexpr (EPI != EPI_QKV) {
                if constexpr (LOM == 1 || LOM == 2) {              // split-fp16 consumers read [hi | lo] (or [hi | hi8 | lo8], gemm.h lo8)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if constexpr (LOM == 2) {
                            if (!CHECK || ok[q]) lo8_store2(dst, off[q] - pcol, pcol, p.lo_off, v0[q], v1[q], lo8_shi, lo8_slo);
                        } else {
                            f16x2 o;
                            o[0] = (f16)(v0[q] - (float)(f16)v0[q]); o[1] = (f16)(v1[q] - (float)(f16)v1[q]);
                            if (!CHECK || ok[q]) *(f16x2 *)(dst + off[q] + p.lo_off) = o;
                        }
                    }
                }
            }
        }
    }
}

// ---- round 4: the same epilogue through buffer resources (EPI_STD, EPI_QKV) ---------------------------------------------------------
// profiles/r04c_gemm8_phase_cycles.log: of a 52.5k-cycle fc1 tile the flat epilogue above takes 13.2k cycles to ISSUE (16.9k with GELU),
// 64 bare `buffer_store_dword` of the same bytes 4.4k.  The difference is address arithmetic: the flat form spends a 64-bit multiply-add
// and shift-add per access and keeps eight 64-bit offsets, eight row indices and eight predicates per pass.  Here an access is
//     resource (tensor base + the wave's first row: SGPRs)  +  lane offset (ONE loop-invariant VGPR per tensor layout)  +  row (scalar soffset)
// so a pass computes no addresses at all.  Rows >= M / columns >= N (CHECK builds only) get an out-of-range lane offset: such stores are
// dropped and such loads return 0.  Arithmetic, rounding and store order are those of direct_epilogue_f16_impl: the outputs are the same bits.
struct BufT {                                   // one tensor as the epilogue addresses it
    __amdgpu_buffer_rsrc_t rs;
    unsigned v;                                 // this lane's byte offset from the wave's first row
};
#define PB_POISON 0xFFFFFF00u
__device__ __forceinline__ BufT buf_of(const void *base, int64_t first_row_bytes, unsigned lane_bytes) {
    BufT t;
    t.rs = make_rsrc((const char *)base + first_row_bytes, 0x80000000u);
    t.v = lane_bytes;
    return t;
}
__device__ __forceinline__ f16x2 bl_h2(const BufT &t, unsigned v, int soff) {
    return __builtin_bit_cast(f16x2, __builtin_amdgcn_raw_buffer_load_b32(t.rs, (int)v, soff, 0));
}
__device__ __forceinline__ f32x2 bl_f2(const BufT &t, unsigned v, int soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(t.rs, (int)v, soff, 0));
}
__device__ __forceinline__ void bs_h2(const BufT &t, unsigned v, int soff, f16x2 x) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), t.rs, (int)v, soff, 0);
}
__device__ __forceinline__ void bs_f2(const BufT &t, unsigned v, int soff, f32x2 x) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), t.rs, (int)v, soff, 0);
}
__device__ __forceinline__ void bs_u16(const BufT &t, unsigned v, int soff, unsigned short x) {
    __builtin_amdgcn_raw_buffer_store_b16(x, t.rs, (int)v, soff, 0);
}
__device__ __forceinline__ unsigned short bl_u16(const BufT &t, unsigned v, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b16(t.rs, (int)v, soff, 0);
}

// FAST >= 0: the activation is the compile-time constant FAST and the launch has no skip tensors, no ReLU'd copy and no pre-ReLU - the
// pass loop is then straight-line code.  With the switches read at run time (FAST = -1) every 8-store pass hops through the uniform
// branches of all the variants, ~1000 instructions apart in a kernel whose code is several times the instruction cache: measured as 7.0k
// cycles for the 64 stores of a plain fp16 tile against 2.4k for the same stores in one run (profiles/r04d / r04e_gemm8_phase_cycles.log).
template <int EPI, int TM, bool CHECK, int LOM, int FAST = -1>
__device__ __forceinline__ void direct_epilogue_buf(const GemmArgs &p, f32x16 (&acc)[TM][2], int wave_m0, int wave_n0, int lane) {
    static_assert(EPI == EPI_STD || EPI == EPI_QKV, "buffer-addressed direct epilogue: EPI_STD / EPI_QKV");
    const int li = lane & 31, lh = lane >> 5;
    const int n = wave_n0 + 2 * li;
    const bool nok = !CHECK || n < p.N;
    const int nc = nok ? n : 0;
    float b0 = 0.f, b1 = 0.f;
    if (p.bias) { b0 = p.bias[nc]; b1 = p.bias[nc + 1]; }
    const int ldo = (int)p.ldo;
    const int64_t row0 = (int64_t)wave_m0 * ldo * 2;                 // byte offset of the wave's first row in an [M, ldo] fp16 tensor
    const unsigned vrow = (unsigned)(4 * lh * ldo * 2);              // the lane's row (4 lh) in such a tensor
    const unsigned v16 = vrow + (unsigned)(nc * 2);                  // + its column pair
    // EPI_QKV: q / k rows live at [b, head, t, 64]; a wave's 128 rows cross a batch boundary at most once and 8-aligned row groups never do
    // (ntp % 8 == 0), so (b, t) of a row is a scalar and the lane only adds its half's 4 rows
    int qk_t00 = 0, qk_b0 = 0;
    float qs = 1.f;
    BufT tq{};
    if constexpr (EPI == EPI_QKV) {
        const int which = wave_n0 / p.D, hn = wave_n0 - which * p.D;
        qs = which == 0 ? p.qscale : 1.f;
        qk_b0 = wave_m0 / p.ntp; qk_t00 = wave_m0 - qk_b0 * p.ntp;
        tq = buf_of(which == 0 ? p.q : p.k, (int64_t)(hn >> 6) * p.ntp * 64 * 2, (unsigned)((4 * lh * 64 + 2 * li) * 2));
    }
    // lo8 (LOM 2): per pixel [hi fp16 (C) | hi8 (C bytes) | lo8 (C bytes)], C = lo_off; this lane's channel pair starts at byte 2 C + nc / 3 C + nc
    const unsigned v8hi = vrow + (unsigned)(2 * p.lo_off + nc), v8lo = v8hi + (unsigned)p.lo_off;
    const unsigned vlo16 = v16 + (unsigned)(p.lo_off * 2);          // LOM 1: the fp16 residual part
    const unsigned vo8 = vrow + (unsigned)(p.o8_off + nc);          // LOM 3: fp8 copy behind the row's fp16 part
    const float lo8_shi = __builtin_ldexpf(1.f, p.lo8_pa), lo8_slo = __builtin_ldexpf(1.f, p.lo8_pa + 12), lo8_inv = __builtin_ldexpf(1.f, -(p.lo8_pa + 12));
    // FAST = act | add1 << 4 | add2 << 5 | out2 << 6 | pre_relu << 7 (the dispatcher's key): every switch is a compile-time constant
    const int act = FAST >= 0 ? (FAST & 15) : p.act;
    const bool gru_r = EPI == EPI_STD && act == ACT_GRU_ZR && wave_n0 >= 128;       // a wave's 64 columns are all z or all r
    // Everything a pass LOADS (skip tensors, GRU state and gate) is fetched by one lambda, and the straight-line copies (FAST >= 0) fetch pass
    // th + 1 BEFORE pass th stores: loads and stores retire on one in-order counter, so loads issued behind a pass's 8-32 stores wait for those
    // stores to reach memory first - eight serialised round trips per tile.  Issued in front of them the compiler's own counted wait
    // (vmcnt = the stores issued since) lets them return while the stores drain.
    struct Pre { f16x2 a1[8], l1[8], a2[8], l2[8]; unsigned short b1[8], b2[8]; f32x2 h[8]; f16x2 z[8]; };
    // (fp16 builds only: with the e4m3 residual parts a pass of the skip-tensor variants already loads 32 registers' worth - two sets of them
    // next to the 128 accumulators spill 260-290 VGPRs)
    // measured (r04n, same box, two builds): no difference - update-block kernel 32.94 ms per step either way - so the prefetch is off by default
    // (its registers would cost the 128 x 128 tile its second workgroup per CU); -DPB_EPI_PIPE builds it for the 256 x 256 tile
#ifdef PB_EPI_PIPE
    constexpr bool PIPE = FAST >= 0 && (LOM == 0 || LOM == 1) && TM == 4;
#else
    constexpr bool PIPE = false;
#endif
    const bool has_add1 = EPI == EPI_STD && (FAST >= 0 ? ((FAST >> 4) & 1) != 0 : p.add1 != nullptr);
    const bool has_add2 = EPI == EPI_STD && (FAST >= 0 ? ((FAST >> 5) & 1) != 0 : p.add2 != nullptr);
    const unsigned vh_r = (unsigned)((4 * lh * 128 + (nc - 128)) * 4), vh_q = (unsigned)((4 * lh * 128 + nc) * 4), vz_q = (unsigned)((4 * lh * 256 + nc) * 2);
    auto rowinfo = [&](int th, int (&rowc)[8], unsigned (&pz)[8]) {
        const int tm = th >> 1, r0 = (th & 1) * 8;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = r0 + q;
            rowc[q] = tm * 32 + (r & 3) + 8 * (r >> 2);
            pz[q] = (!CHECK || (nok && wave_m0 + rowc[q] + 4 * lh < p.M)) ? 0u : PB_POISON;
        }
    };
    auto fetch = [&](int th, Pre &x, int parts) {            // parts: 1 add1, 2 add2, 4 the GRU's state / gate
        if constexpr (EPI == EPI_STD) {
            int rowc[8];
            unsigned pz[8];
            rowinfo(th, rowc, pz);
            if ((parts & 1) && has_add1) {
                const BufT ta = buf_of(p.add1, row0, 0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    x.a1[q] = bl_h2(ta, v16 | pz[q], rowc[q] * ldo * 2);
                    if constexpr (LOM == 2) x.b1[q] = bl_u16(ta, v8lo | pz[q], rowc[q] * ldo * 2);
                    if constexpr (LOM == 1) x.l1[q] = bl_h2(ta, vlo16 | pz[q], rowc[q] * ldo * 2);
                }
            }
            if ((parts & 2) && has_add2) {
                const BufT ta = buf_of(p.add2, row0, 0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    x.a2[q] = bl_h2(ta, v16 | pz[q], rowc[q] * ldo * 2);
                    if constexpr (LOM == 2) x.b2[q] = bl_u16(ta, v8lo | pz[q], rowc[q] * ldo * 2);
                    if constexpr (LOM == 1) x.l2[q] = bl_h2(ta, vlo16 | pz[q], rowc[q] * ldo * 2);
                }
            }
            if ((parts & 4) && gru_r) {
                const BufT th_ = buf_of(p.gru_h, (int64_t)wave_m0 * 128 * 4, 0);
#pragma unroll
                for (int q = 0; q < 8; ++q) x.h[q] = bl_f2(th_, vh_r | pz[q], rowc[q] * 128 * 4);
            }
            if ((parts & 4) && act == ACT_GRU_Q) {
                const BufT th_ = buf_of(p.gru_h, (int64_t)wave_m0 * 128 * 4, 0);
                const BufT tz = buf_of(p.gru_z, (int64_t)wave_m0 * 256 * 2, 0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    x.h[q] = bl_f2(th_, vh_q | pz[q], rowc[q] * 128 * 4);
                    x.z[q] = bl_h2(tz, vz_q | pz[q], rowc[q] * 256 * 2);
                }
            }
        }
    };
    Pre nxt;
    if constexpr (PIPE) fetch(0, nxt, 7);
#pragma unroll
    for (int th = 0; th < TM * 2; ++th) {                   // 8 accumulator registers (= 16 rows) per pass
        const int tm = th >> 1, r0 = (th & 1) * 8;
        // tile-local row of register q (a compile-time constant) and, CHECK builds, the poisoned lane offsets of rows >= M
        int rowc[8];
        unsigned pz[8];                                     // 0 or PB_POISON: OR-ed into every lane offset of row q
        rowinfo(th, rowc, pz);
        Pre cur;
        if constexpr (PIPE) {
            cur = nxt;
            if (th + 1 < TM * 2) fetch(th + 1, nxt, 7);
        }
        float v0[8], v1[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { v0[q] = acc[tm][0][r0 + q] + b0; v1[q] = acc[tm][1][r0 + q] + b1; }
        if constexpr (EPI == EPI_STD) {
            if (FAST >= 0 ? ((FAST >> 7) & 1) != 0 : p.pre_relu != 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
            }
            if constexpr (!PIPE) fetch(th, cur, 1);          // (not pipelined: each part is loaded where it is consumed, one register set at a time)
            if (has_add1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] += (float)cur.a1[q][0]; v1[q] += (float)cur.a1[q][1]; }
                if constexpr (LOM == 2) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const f32x2 l = __builtin_amdgcn_cvt_pk_f32_fp8((int)cur.b1[q], false) * lo8_inv; v0[q] += l[0]; v1[q] += l[1]; }
                } else if constexpr (LOM == 1) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { v0[q] += (float)cur.l1[q][0]; v1[q] += (float)cur.l1[q][1]; }
                }
            }
            if constexpr (!PIPE) fetch(th, cur, 2);
            if (has_add2) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] += (float)cur.a2[q][0]; v1[q] += (float)cur.a2[q][1]; }
                if constexpr (LOM == 2) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const f32x2 l = __builtin_amdgcn_cvt_pk_f32_fp8((int)cur.b2[q], false) * lo8_inv; v0[q] += l[0]; v1[q] += l[1]; }
                } else if constexpr (LOM == 1) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { v0[q] += (float)cur.l2[q][0]; v1[q] += (float)cur.l2[q][1]; }
                }
            }
            if (FAST >= 0 ? ((FAST >> 6) & 1) != 0 : p.out2 != nullptr) {
                const BufT t2 = buf_of(p.out2, row0, 0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    f16x2 o2;
                    o2[0] = (f16)fmaxf(v0[q], 0.f); o2[1] = (f16)fmaxf(v1[q], 0.f);
                    bs_h2(t2, v16 | pz[q], rowc[q] * ldo * 2, o2);
                    if constexpr (LOM == 3) bs_u16(t2, vo8 | pz[q], rowc[q] * ldo * 2, pb_fp8x2((float)o2[0] * p.o8_scale, (float)o2[1] * p.o8_scale));
                }
                if constexpr (LOM == 1 || LOM == 2) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float r0f = fmaxf(v0[q], 0.f), r1f = fmaxf(v1[q], 0.f);
                        if constexpr (LOM == 2) {
                            const f16 h0 = (f16)r0f, h1 = (f16)r1f;
                            bs_u16(t2, v8hi | pz[q], rowc[q] * ldo * 2, pb_fp8x2((float)h0 * lo8_shi, (float)h1 * lo8_shi));
                            bs_u16(t2, v8lo | pz[q], rowc[q] * ldo * 2, pb_fp8x2((r0f - (float)h0) * lo8_slo, (r1f - (float)h1) * lo8_slo));
                        } else {
                            f16x2 o2;
                            o2[0] = (f16)(r0f - (float)(f16)r0f); o2[1] = (f16)(r1f - (float)(f16)r1f);
                            bs_h2(t2, vlo16 | pz[q], rowc[q] * ldo * 2, o2);
                        }
                    }
                }
            }
            if constexpr (!PIPE) fetch(th, cur, 4);
            if (act == ACT_GELU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) fast_gelu2(v0[q], v1[q]);
            } else if (act == ACT_RELU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
            } else if (act == ACT_SIGMOID) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] = __frcp_rn(1.f + __expf(-v0[q])); v1[q] = __frcp_rn(1.f + __expf(-v1[q])); }
            } else if (act == ACT_TANH) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] = fast_tanh(v0[q]); v1[q] = fast_tanh(v1[q]); }
            } else if (act == ACT_GRU_ZR) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] = __frcp_rn(1.f + __expf(-v0[q])); v1[q] = __frcp_rn(1.f + __expf(-v1[q])); }
                if (gru_r) {                     // r columns: r * h -> gru_rh (the z columns fall through to the plain store below)
                    const BufT tr = buf_of(p.gru_rh, (int64_t)wave_m0 * p.gru_ld * 2, 0);
                    const unsigned vr = (unsigned)((4 * lh * p.gru_ld + (nc - 128)) * 2);
                    const unsigned vr8 = (unsigned)(4 * lh * p.gru_ld * 2 + p.o8_off + (nc - 128));
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        f16x2 o;
                        o[0] = (f16)(v0[q] * cur.h[q][0]); o[1] = (f16)(v1[q] * cur.h[q][1]);
                        bs_h2(tr, vr | pz[q], rowc[q] * p.gru_ld * 2, o);
                        if constexpr (LOM == 3)       // fp8 twin of r * h for the q convolution's MX segment (o8_off = byte offset of the copy)
                            bs_u16(tr, vr8 | pz[q], rowc[q] * p.gru_ld * 2, pb_fp8x2((float)o[0] * p.o8_scale, (float)o[1] * p.o8_scale));
                    }
                    continue;
                }
            } else if (act == ACT_GRU_Q) {
                const BufT th_ = buf_of(p.gru_h, (int64_t)wave_m0 * 128 * 4, 0);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float z0 = (float)cur.z[q][0], z1 = (float)cur.z[q][1];
                    f32x2 hn;
                    hn[0] = (1.f - z0) * cur.h[q][0] + z0 * fast_tanh(v0[q]);
                    hn[1] = (1.f - z1) * cur.h[q][1] + z1 * fast_tanh(v1[q]);
                    bs_f2(th_, vh_q | pz[q], rowc[q] * 128 * 4, hn);
                    v0[q] = hn[0]; v1[q] = hn[1];
                }
            }
        }
        if (EPI != EPI_STD || p.out) {
            BufT td;
            if constexpr (EPI == EPI_QKV) td = tq; else td = buf_of(p.out, row0, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                f16x2 o;
                o[0] = (f16)(v0[q] * qs); o[1] = (f16)(v1[q] * qs);
                if constexpr (EPI == EPI_QKV) {
                    int t = qk_t00 + rowc[q], b = qk_b0;
                    if (t >= p.ntp) { t -= p.ntp; b += 1; }
                    bs_h2(td, td.v | pz[q], (b * p.heads * p.ntp + t) * 64 * 2, o);
                } else {
                    bs_h2(td, v16 | pz[q], rowc[q] * ldo * 2, o);
                }
            }
            if constexpr (EPI == EPI_STD && LOM == 3) {
                if (act != ACT_GRU_ZR) {       // fp8 copy for a consumer's MX correction segment (gemm.h o8_off); z itself has none
#pragma unroll
                    for (int q = 0; q < 8; ++q) bs_u16(td, vo8 | pz[q], rowc[q] * ldo * 2, pb_fp8x2(v0[q] * p.o8_scale, v1[q] * p.o8_scale));
                }
            }
            if constexpr (EPI == EPI_STD && (LOM == 1 || LOM == 2)) {       // split-fp16 consumers read [hi | lo] (or [hi | hi8 | lo8], gemm.h lo8)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if constexpr (LOM == 2) {
                        const f16 h0 = (f16)v0[q], h1 = (f16)v1[q];
                        bs_u16(td, v8hi | pz[q], rowc[q] * ldo * 2, pb_fp8x2((float)h0 * lo8_shi, (float)h1 * lo8_shi));
                        bs_u16(td, v8lo | pz[q], rowc[q] * ldo * 2, pb_fp8x2((v0[q] - (float)h0) * lo8_slo, (v1[q] - (float)h1) * lo8_slo));
                    } else {
                        f16x2 o;
                        o[0] = (f16)(v0[q] - (float)(f16)v0[q]); o[1] = (f16)(v1[q] - (float)(f16)v1[q]);
                        bs_h2(td, vlo16 | pz[q], rowc[q] * ldo * 2, o);
                    }
                }
            }
        }
    }
}

// ---- round 5: the pixel-shuffle epilogue through a buffer resource --------------------------------------------------------------------
// Row m = (b, y, x) of the [B, ps_h, ps_w] grid lands at output pixel P(m) = s m + s (s - 1) ps_w Y(m) + (dy ps_w s + dx), Y = m / ps_w
// (image rows run on across the batch: b ps_h + y).  The flat epilogue divides twice per row (64 rows per lane and tile: ~5k instructions
// of integer division) and carries 64-bit offsets; the stem of the flow band's encoder (K = 9 tiles, the epilogue IS the kernel) ran at
// 0.7 TB/s of output.  Here the wave divides ONCE (its first row), the resource starts at that row's pixel, a row is
//     scalar offset (s x tile-local row)  +  lane offset (its half's 4 rows, its channel pair)  +  wraps x (s (s - 1) ps_w pixels)
// where wraps = how many grid rows the lane's row lies past the wave's first one: sum over k of (x0 + d >= k ps_w), k <= TM (ps_w >= 32).
// Arithmetic, rounding and store order are those of direct_epilogue_f16_impl: the same bits.
template <int TM, bool CHECK, int LOM>
__device__ __forceinline__ void pixshuf_epilogue_buf(const GemmArgs &p, f32x16 (&acc)[TM][2], int wave_m0, int wave_n0, int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const int n = wave_n0 + 2 * li;
    const bool nok = !CHECK || n < p.N;
    const int tap = wave_n0 / p.ps_co;                           // a wave's 64 columns never straddle a tap (ps_co % 64 == 0)
    const int co = wave_n0 - tap * p.ps_co + 2 * li;
    const int tap_dy = tap / p.ps_s, tap_dx = tap - tap_dy * p.ps_s;
    float b0 = 0.f, b1 = 0.f;
    if (p.bias) { const int bi = nok ? co : 0; b0 = p.bias[bi]; b1 = p.bias[bi + 1]; }
    const int s = p.ps_s, psw = p.ps_w;
    const int pb = (int)p.ldo * 2;                               // bytes per output pixel
    const int x0 = wave_m0 - (wave_m0 / psw) * psw;
    const int64_t pix0 = pixshuf_first_pixel(wave_m0, s, psw, tap_dy, tap_dx);
    const BufT td = buf_of(p.out, pix0 * pb, 0);
    const unsigned wrapb = (unsigned)(s * (s - 1) * psw) * (unsigned)pb;
    const unsigned vrow = (unsigned)(4 * lh * s * pb);
    const unsigned v16 = vrow + (unsigned)(co * 2), vlo16 = v16 + (unsigned)(p.lo_off * 2);
    const unsigned v8hi = vrow + (unsigned)(2 * p.lo_off + co), v8lo = v8hi + (unsigned)p.lo_off;
    const float lo8_shi = __builtin_ldexpf(1.f, p.lo8_pa), lo8_slo = __builtin_ldexpf(1.f, p.lo8_pa + 12);
    const int xl = x0 + 4 * lh;
    const bool relu = p.act == ACT_RELU;
#pragma unroll
    for (int th = 0; th < TM * 2; ++th) {
        const int tm = th >> 1, r0 = (th & 1) * 8;
        float v0[8], v1[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { v0[q] = acc[tm][0][r0 + q] + b0; v1[q] = acc[tm][1][r0 + q] + b1; }
        if (relu) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
        }
        unsigned wv[8], pz[8];
        int so[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = r0 + q;
            const int rowc = tm * 32 + (r & 3) + 8 * (r >> 2);
            wv[q] = (unsigned)pixshuf_wraps<TM>(xl + rowc, psw) * wrapb;
            pz[q] = (!CHECK || (nok && wave_m0 + rowc + 4 * lh < p.M)) ? 0u : PB_POISON;       // OR-ed in: an add would wrap around
            so[q] = rowc * s * pb;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f16x2 o;
            o[0] = (f16)v0[q]; o[1] = (f16)v1[q];
            bs_h2(td, (v16 + wv[q]) | pz[q], so[q], o);
        }
        if constexpr (LOM == 1 || LOM == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if constexpr (LOM == 2) {
                    const f16 h0 = (f16)v0[q], h1 = (f16)v1[q];
                    bs_u16(td, (v8hi + wv[q]) | pz[q], so[q], pb_fp8x2((float)h0 * lo8_shi, (float)h1 * lo8_shi));
                    bs_u16(td, (v8lo + wv[q]) | pz[q], so[q], pb_fp8x2((v0[q] - (float)h0) * lo8_slo, (v1[q] - (float)h1) * lo8_slo));
                } else {
                    f16x2 o;
                    o[0] = (f16)(v0[q] - (float)(f16)v0[q]); o[1] = (f16)(v1[q] - (float)(f16)v1[q]);
                    bs_h2(td, (vlo16 + wv[q]) | pz[q], so[q], o);
                }
            }
        }
    }
}

template <int EPI, int TM, bool CHECK, int LOM>
__device__ __forceinline__ void direct_epilogue_any(const GemmArgs &p, f32x16 (&acc)[TM][2], int wave_m0, int wave_n0, int lane) {
#ifdef PB_EPI_FLAT_STD      // A/B builds only (make EXTRA=-DPB_EPI_FLAT_STD): the round-3 flat-addressed epilogue for EPI_STD
    if constexpr (EPI == EPI_STD) { direct_epilogue_f16_impl<EPI, TM, CHECK, LOM>(p, acc, wave_m0, wave_n0, lane); return; }
#endif
    // fp16-residual split maps ([hi | lo] pairs: the mask band) on the ping-pong kernel keep the flat-addressed epilogue: measured on the
    // band's 1 x 1 ResNet convolutions (K = 64-512: the epilogue IS the kernel) 27.4 ms per 32 frames against 40.6 with the buffer-addressed one
    // (profiles/r04j_mask_epilogue_ab.txt); every other format and the 128 x 128 tile measured equal or faster on the buffer path
    if constexpr (EPI == EPI_STD && LOM == 1 && TM == 4) { direct_epilogue_f16_impl<EPI, TM, CHECK, LOM>(p, acc, wave_m0, wave_n0, lane); return; }
    if constexpr (EPI == EPI_STD) {
        // interior tiles of the common launches take a straight-line copy of the epilogue (direct_epilogue_buf FAST): the plain / ReLU / GELU
        // outputs of the linears and convolutions, the two GRU gates, and the DPT head's skip / ReLU'd-copy combinations
        if (!CHECK) {
            const int key = p.act | (p.add1 ? 16 : 0) | (p.add2 ? 32 : 0) | (p.out2 ? 64 : 0) | (p.pre_relu ? 128 : 0);
#define PB_FAST_CASE(K) case K: direct_epilogue_buf<EPI, TM, false, LOM, K>(p, acc, wave_m0, wave_n0, lane); return;
            switch (key) {
                PB_FAST_CASE(ACT_NONE) PB_FAST_CASE(ACT_RELU) PB_FAST_CASE(ACT_GELU) PB_FAST_CASE(ACT_GRU_ZR) PB_FAST_CASE(ACT_GRU_Q)
                PB_FAST_CASE(ACT_NONE | 64) PB_FAST_CASE(ACT_NONE | 16) PB_FAST_CASE(ACT_NONE | 16 | 32 | 64)
                PB_FAST_CASE(ACT_GRU_ZR | 16 | 32) PB_FAST_CASE(ACT_GRU_Q | 16 | 32) PB_FAST_CASE(ACT_RELU | 16 | 128) PB_FAST_CASE(ACT_RELU | 64) PB_FAST_CASE(ACT_RELU | 16)
                default: break;
            }
#undef PB_FAST_CASE
        }
        direct_epilogue_buf<EPI, TM, CHECK, LOM>(p, acc, wave_m0, wave_n0, lane);
    } else if constexpr (EPI == EPI_QKV) direct_epilogue_buf<EPI, TM, CHECK, LOM>(p, acc, wave_m0, wave_n0, lane);
    else if constexpr (EPI == EPI_PIXSHUF && LOM != 3) {
        // (ps_buf: set by the launcher when ps_w >= 32 and TM x 32 rows of output pixels + the wraps stay inside a 2 GB window)
        if (p.ps_buf) pixshuf_epilogue_buf<TM, CHECK, LOM>(p, acc, wave_m0, wave_n0, lane);
        else direct_epilogue_f16_impl<EPI, TM, CHECK, LOM>(p, acc, wave_m0, wave_n0, lane);
    }
    else direct_epilogue_f16_impl<EPI, TM, CHECK, LOM>(p, acc, wave_m0, wave_n0, lane);
}

template <int EPI, int TM, bool MX>
__device__ __forceinline__ void direct_epilogue_f16(const GemmArgs &p, f32x16 (&acc)[TM][2], int wave_m0, int wave_n0, int lane) {
    // interior tiles (the common case) store without per-lane predicates: each predicate costs an exec-mask branch
    // split-fp16 outputs (p.lo_off) take their own copy of the code: the plain path keeps its register allocation
    if constexpr (EPI == EPI_STD || EPI == EPI_PIXSHUF) {
        if (p.lo_off) {
            // MX builds serve the e4m3-residual maps (lo8), fp16-only builds the fp16-residual maps
            const bool inner = wave_m0 + TM * 32 <= p.M && wave_n0 + 64 <= p.N;
            if (inner) direct_epilogue_any<EPI, TM, false, MX ? 2 : 1>(p, acc, wave_m0, wave_n0, lane);
            else direct_epilogue_any<EPI, TM, true, MX ? 2 : 1>(p, acc, wave_m0, wave_n0, lane);
            return;
        }
    }
    if constexpr (MX && EPI == EPI_STD) {
        if (p.o8_off) {
            if (wave_m0 + TM * 32 <= p.M && wave_n0 + 64 <= p.N) direct_epilogue_any<EPI, TM, false, 3>(p, acc, wave_m0, wave_n0, lane);
            else direct_epilogue_any<EPI, TM, true, 3>(p, acc, wave_m0, wave_n0, lane);
            return;
        }
    }
    if (wave_m0 + TM * 32 <= p.M && wave_n0 + 64 <= p.N) direct_epilogue_any<EPI, TM, false, 0>(p, acc, wave_m0, wave_n0, lane);
    else direct_epilogue_any<EPI, TM, true, 0>(p, acc, wave_m0, wave_n0, lane);
}

// Accumulators -> per-wave LDS patch -> 8-column chunks -> fused store.  `smem` must be free of live
// staging data (callers barrier first).  TM x TN = 32x32 MFMA tiles per wave.
template <int EPI, int TM, int TN, bool MX>
__device__ __forceinline__ void run_epilogue(const GemmArgs &p, f32x16 (&acc)[TM][TN], char *smem, int wave, int lane,
                                             int wave_m0, int wave_n0, int n0) {
    if constexpr (TN == 3) {              // 128 x 96 tile: an interleaved pair of column blocks + a single one
        static_assert(EPI == EPI_STD, "the 96-wide tile is built for EPI_STD");
        f32x16 pair[TM][2];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) { pair[tm][0] = acc[tm][0]; pair[tm][1] = acc[tm][1]; }
        direct_epilogue_f16<EPI, TM, MX>(p, pair, wave_m0, wave_n0, lane);
        single_col_epilogue<TM, MX>(p, acc, wave_m0, wave_n0 + 64, lane);
        return;
    }
    constexpr int ES = TN * 32 + 4;
    constexpr int EPIB = (EPI == EPI_QKV) ? TN * 32 * 36 * 4 : 32 * ES * 4;
    constexpr bool IL = epi_interleaved<EPI, TN>();
    const int li = lane & 31, lh = lane >> 5;
    float *es = (float *)(smem + wave * EPIB);

    if constexpr (EPI == EPI_QKV) {
        if (n0 >= 2 * p.D) {
            // V third: transpose through LDS so that stores run along the token axis of Vt.
            float bb[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bb[tn] = p.bias[wave_n0 + (IL ? 2 * li + tn : tn * 32 + li)];
#pragma unroll
            for (int tmi = 0; tmi < TM; ++tmi) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 w4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) w4[j] = acc[tmi][tn][g * 4 + j] + bb[tn];
                        *(f32x4 *)(es + (IL ? 2 * li + tn : tn * 32 + li) * 36 + 8 * g + 4 * lh) = w4;
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int it = 0; it < TN * 2; ++it) {
                    const int item = it * 64 + lane;
                    const int nr = item >> 2, mc = item & 3;
                    const f32x4 x0 = *(const f32x4 *)(es + nr * 36 + mc * 8);
                    const f32x4 x1 = *(const f32x4 *)(es + nr * 36 + mc * 8 + 4);
                    const int m = wave_m0 + tmi * 32 + mc * 8;
                    const int n = wave_n0 + nr - 2 * p.D;
                    if (m < p.M) {
                        const int b = m / p.ntp, t = m - b * p.ntp;
                        f16x8 r;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { r[j] = (f16)x0[j]; r[4 + j] = (f16)x1[j]; }
                        *(f16x8 *)(p.vt + (((int64_t)b * p.heads + (n >> 6)) * 64 + (n & 63)) * p.ntp + t) = r;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            return;
        }
    }

    if constexpr (IL && TN == 2) {
        const bool direct = EPI != EPI_PIXSHUF || (p.ps_co & 63) == 0;
        if (direct) {
            direct_epilogue_f16<EPI, TM, MX>(p, acc, wave_m0, wave_n0, lane);
            return;
        }
    }
    constexpr int CPR = TN * 4;                             // 8-column chunks per patch row
    constexpr int NIT = 32 * CPR / 64;                      // rows per lane per pass
    constexpr int RSTEP = 64 / CPR;
    const int ch = lane % CPR, row0 = lane / CPR;
    const int n = wave_n0 + ch * 8;
    const bool nok = n < p.N;
    float cb[8], cg[8];
    epi_cols<EPI>(p, n, nok, cb, cg);

#pragma unroll
    for (int tmi = 0; tmi < TM; ++tmi) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                es[((r & 3) + 8 * (r >> 2) + 4 * lh) * ES + (IL ? 2 * li + tn : tn * 32 + li)] = acc[tmi][tn][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float v[NIT][8];
        EpiAux aux[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RSTEP + row0;
            const f32x4 x0 = *(const f32x4 *)(es + row * ES + ch * 8);
            const f32x4 x1 = *(const f32x4 *)(es + row * ES + ch * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[it][j] = x0[j]; v[it][4 + j] = x1[j]; }
            const int m = wave_m0 + tmi * 32 + row;
            if (m < p.M && nok) epi_prefetch<EPI>(p, m, n, aux[it]);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int m = wave_m0 + tmi * 32 + it * RSTEP + row0;
            if constexpr (EPI == EPI_HEAD) {
                // N == 32: the 4 lanes of a row hold its 32 channels
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s += fmaxf(v[it][j] + cb[j], 0.f) * cg[j];
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                if (ch == 0 && m < p.M) p.depth[m] = fmaxf(s + p.b2, 0.f);
            } else {
                if (m < p.M && nok) epi_finish<EPI>(p, m, n, v[it], aux[it], cb, cg);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

#define PB_BAR_GENERIC()                         \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        asm volatile("s_barrier" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

__device__ __forceinline__ void vm_wait_halftiles(int n) {
    if (n >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// NS LDS stages (2 or 3).  With 2 the DMA of tile kt+1 has one tile of MFMAs (~0.5k cycles) to land; with 3 the loads run
// two tiles ahead and the wait before the barrier leaves the newest stage in flight (counted vmcnt, never drained).
// MX: the launch has MX-fp8 tiles (gemm.h nk16); the fp16-only build of every kernel keeps the plain K loop
template <int BM, int BN, int WM, int WN, int AMODE, int EPI, bool BUFP = false, int NS = 2, bool MX = false>
__global__ __launch_bounds__(WM *WN * 64) void gemm_kernel(const GemmArgs p) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = BM * 8 / NT, NB = BN * 8 / NT;       // 16-byte chunks per thread per stage
    static_assert(NA >= 1 && NB >= 1, "tile too small for the block");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    stagger_start(p.stagger, BM == 128 ? 512 : 256);

    // ---- tile id with XCD-contiguous remap (bijective for any grid size) ----
    const int tilesN = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    const int swz_all = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    // split-K (gemm.h splitk): the grid holds `sk` slices of every tile, slice-major
    const int sk = p.sk_phase == 1 ? p.splitk : 1;
    const int ntile = sk > 1 ? nwg / sk : nwg;
    const int kslice = sk > 1 ? swz_all / ntile : 0;
    const int swz = swz_all - kslice * ntile;
    const int tile_m = swz / tilesN, tile_n = swz - tile_m * tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread staging addresses ----
    const int cld = p.cLd ? p.cLd : p.cC, padx = p.cPadX >= 0 ? p.cPadX : p.cPad;
    const int srow = tid >> 3;                              // + i * (NT/8)
    const int cg = (tid & 7) ^ ((tid >> 4) & 7);            // swizzled global chunk for this LDS slot
    // CW (the 128 x 96 tile's convolutions): the K walk is a table word per 16-byte CHUNK of a tile (conv_walk.h chunk walk) - a lane stages chunk cg
    // of every tile and reads its own word - so that a K tile may mix taps (the packed-channel K axis of gemm.h Wcw); classic layouts expand into it
    constexpr bool CW = AMODE == A_CONV && BN == 96;
    const f16 *a_ptr[NA];
    // conv: a_msk = tap_mask of the row's pixel (0 for rows >= M), a_pix0 = element offset of its tap (0, 0) inside the image (may be negative)
    unsigned a_msk[NA];
    int a_pix0[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + srow + i * (NT / 8);
        if constexpr (AMODE == A_DENSE) {
            const int mc = m < p.M ? m : p.M - 1;
            a_ptr[i] = p.A + (int64_t)mc * p.lda + cg * 8;
            a_msk[i] = 0u;
            a_pix0[i] = 0;
        } else {
            const int ohw = p.cOH * p.cOW;
            const int b = m / ohw, rem = m - b * ohw;
            const int oy = rem / p.cOW, ox = rem - oy * p.cOW;
            const int iy0 = oy * p.cStride - p.cPad, ix0 = ox * p.cStride - padx;
            a_msk[i] = m < p.M ? tap_mask(iy0, ix0, p.cH, p.cW) : 0u;
            a_pix0[i] = (iy0 * p.cW + ix0) * cld;
            a_ptr[i] = p.A + (int64_t)b * p.cH * p.cW * cld + (CW ? 0 : cg * 8);       // (CW: the chunk's offset comes from its table word)
        }
    }
    const f16 *b_ptr[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
        b_ptr[i] = p.W + (int64_t)(n0 + col_map_tn<TN>(srow + i * (NT / 8), epi_interleaved<EPI, TN>())) * p.K + cg * 8;

    int c_ky = 0, c_kx = 0, c_c0 = 0;                       // conv tap state of the NEXT stage call (!KT builds)
    const int nk = p.K >> 6;
    // this workgroup's K tiles [k0, k1): all of them unless the launch is split along K (only KT / dense builds: the cursor walk starts at tile 0)
    // (the second launch of a split-K GEMM - sk_phase 2 - has none: it starts from the parked slices and runs the epilogue)
    const int k0 = sk > 1 ? (int)((int64_t)kslice * nk / sk) : 0, k1 = sk > 1 ? (int)((int64_t)(kslice + 1) * nk / sk) : (p.sk_phase == 2 ? 0 : nk);
    // KT: the K walk comes from the table in the LDS (conv_ktab_entry).  The 256 x 64 tile keeps the cursor: its two workgroups per CU
    // use the LDS to the last byte.
    constexpr bool KT = AMODE == A_CONV && BN != 64;
    const unsigned *ktab = (const unsigned *)(smem + NS * STAGE);
    unsigned e_nxt = 0;                                     // table word of the next stage call, read one call ahead
    if constexpr (KT) {
        if constexpr (CW) {
            for (int q = tid; q < nk * 8; q += NT)
                ((unsigned *)(smem + NS * STAGE))[q] = p.cwalk ? conv_cw3_word(p.cKW, p.cwTaps, p.cwC, p.cwPad, p.cW, cld, q)
                                                               : conv_ctab_classic(conv_ktab_entry(p, cld, q >> 3), q & 7);
        } else {
            for (int t = tid; t < nk; t += NT) ((unsigned *)(smem + NS * STAGE))[t] = conv_ktab_entry(p, cld, t);
        }
        __syncthreads();
        e_nxt = CW ? ktab[k0 * 8 + cg] : ktab[k0];
    }
    // BUFP: LDS-DMA through the buffer path (see gemm8_kernel); p.bufmode 1 = whole operand, 2 = two-image window (conv)
    __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, 0u), rsW = make_rsrc(p.W, 0u);
    unsigned a_voff[NA], b_voff[NB];
    if constexpr (BUFP) {
        rsW = make_rsrc(p.W, (unsigned)((int64_t)((p.N + 255) / 256 * 256) * p.K * 2));
        int b0 = 0;
        if constexpr (AMODE == A_DENSE) {
            rsA = make_rsrc(p.A, (unsigned)((int64_t)p.M * p.lda * 2));
        } else {
            const int ohw = p.cOH * p.cOW, nimg = p.M / ohw;
            const int64_t img = (int64_t)p.cH * p.cW * cld;
            int cnt = nimg;
            if (p.bufmode == 2) { b0 = m0 / ohw; cnt = nimg - b0 < 2 ? nimg - b0 : 2; }
            rsA = make_rsrc(p.A + b0 * img, (unsigned)(cnt * img * 2));
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if constexpr (AMODE == A_DENSE) a_voff[i] = (unsigned)((a_ptr[i] - p.A) * 2);
            else a_voff[i] = (unsigned)((int64_t)((m0 + srow + i * (NT / 8)) / (p.cOH * p.cOW) - b0) * p.cH * p.cW * cld * 2) + (CW ? 0 : cg * 16) + (unsigned)(a_pix0[i] * 2);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) b_voff[i] = (unsigned)((b_ptr[i] - p.W) * 2);
    }

    auto stage = [&](int buf, int kt) {
        char *sA = smem + buf * STAGE + wave * 1024;
        char *sB = smem + buf * STAGE + A_BYTES + wave * 1024;
        const int ka = (p.kwrap && kt >= p.kwrap) ? kt - p.kwrap : kt;            // split-fp16 segments re-read A (gemm.h)
        const int cs = (p.kwrap && c_c0 >= p.kwrap) ? c_c0 + p.kshift : c_c0;
        // scalar per K tile: the tap's element offset and its two bits of tap_mask - a DMA then costs an AND, a compare, an add and a select
        // (round 4: the per-DMA (iy * W + ix) * ld was a 64-bit mad + a 32-bit multiply, ~70 VALU cycles per DMA, 4 DMAs per 16 MFMAs)
        unsigned tapoff2 = (unsigned)(((c_ky * p.cW + c_kx) * cld + cs) * 2), tapsel = (1u << c_ky) | (256u << c_kx);
        if constexpr (CW) {                         // per lane: this lane's chunk of tile kt
            const unsigned e = e_nxt;
            e_nxt = ktab[(kt + 1 < nk ? kt + 1 : nk - 1) * 8 + cg];
            tapoff2 = ktab_bytes(e); tapsel = ktab_sel_chunk(e);
        } else if constexpr (KT) {
            const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)e_nxt);
            e_nxt = ktab[kt + 1 < nk ? kt + 1 : nk - 1];
            tapoff2 = ktab_bytes(e); tapsel = ktab_sel(e);
        }
#ifdef PB_DIAG
        // PB_GEMM_ABL=4 (timing only, wrong results): the A DMAs of the taps with kx != 0 fetch nothing (an impossible tap test sends them out of
        // range: zeros arrive in the LDS without an L2 / HBM access) - the fetch side of what a staging that shares one (rows + 2)-pixel window
        // between the kx taps of a (slice, ky) would save (EXPERIMENTS.md 6.5).  Scalar select: no register, no branch.
        // PB_GEMM_ABL=8: the same taps read a 32 KB window at the start of the operand instead (cache hits; random data, so the MFMAs keep
        // their switching activity where the zeros of mode 4 lower it).  PB_GEMM_ABL=16: their DMAs are not issued at all (the MFMAs multiply
        // what an earlier K tile left in the LDS stage: random data as well) - the upper bound of a shared window, LDS writes included
        if constexpr (AMODE == A_CONV) tapsel = ((p.ablate & 4) && (tapsel >> 8) != 1u) ? 0xFFFFFFFFu : tapsel;
        const bool abl_near = AMODE == A_CONV && (p.ablate & 8) && (tapsel >> 8) != 1u;
#endif
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if constexpr (AMODE == A_DENSE) {
                if constexpr (BUFP) glds16_buf(rsA, (int)a_voff[i], ka * 128, sA + i * (NT * 16));
                else glds16(a_ptr[i] + ka * 64, sA + i * (NT * 16));
            } else {
                const bool ok = (a_msk[i] & tapsel) == tapsel;
                if constexpr (BUFP) {      // branch-free: an out-of-range offset makes the load return zeros
#ifdef PB_DIAG
                    if (abl_near) { glds16_buf(rsA, (int)((unsigned)tid * 16u + (tapoff2 & 0x7000u)), 0, sA + i * (NT * 16)); continue; }
                    if ((p.ablate & 16) && (tapsel >> 8) != 1u) continue;
#endif
                    glds16_buf(rsA, (int)(ok ? a_voff[i] + tapoff2 : 0xFFFFFF00u), 0, sA + i * (NT * 16));
                } else {
                    glds16(ok ? a_ptr[i] + (a_pix0[i] + (int)(tapoff2 >> 1)) : p.zero, sA + i * (NT * 16));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if constexpr (BUFP) glds16_buf(rsW, (int)b_voff[i], kt * 128, sB + i * (NT * 16));
            else glds16(b_ptr[i] + kt * 64, sB + i * (NT * 16));
        }
        if constexpr (AMODE == A_CONV && !KT) {
            // one branch-free step for both K orders (gemm.h cTapInner): tap-major walks (ky, kx, c), slice-major (c, ky, kx)
            const int ti = p.cTapInner;
            int c1 = c_c0 + (ti ? 0 : 64);
            const int wc = (!ti && c1 >= p.cC) ? 1 : 0;
            c1 = wc ? 0 : c1;
            int x1 = c_kx + ti + wc;
            const int wx = x1 == p.cKW ? 1 : 0;
            c_kx = wx ? 0 : x1;
            const int y1 = c_ky + wx;
            const int wy = (ti && y1 == p.cKH) ? 1 : 0;
            c_ky = wy ? 0 : y1;
            c_c0 = c1 + 64 * wy;
        }
    };

    // ---- fragment read offsets ----
    const int li = lane & 31, lh = lane >> 5;
    const int fsw = (li >> 1) & 7;
    int a_off[TM], b_off[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) a_off[t] = ((wm * TM + t) * 32 + li) * 128;
#pragma unroll
    for (int t = 0; t < TN; ++t) b_off[t] = A_BYTES + ((wn * TN + t) * 32 + li) * 128;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if constexpr (EPI == EPI_RESID) {
        if (p.sk_phase != 1) resid_io<TM, TN, false>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, lane);      // (split-K: the reducing launch adds the residual rows)
    }
    if (p.sk_phase == 2) {          // split-K, second launch: the slices of this tile, in slice order
        constexpr int NACC = TM * TN * 16;
        for (int sl = 0; sl < p.splitk; ++sl) {
            const float *part = p.sk_ws + ((int64_t)(sl * ntile + swz) * NACC) * NT + tid;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += part[(int64_t)((i * TN + j) * 16 + r) * NT];
        }
    }

    // K loop: one barrier per K tile; inside a tile the fragments of k-step ks+1 are read from LDS while the
    // MFMAs of k-step ks run (register double buffer), and the next tile's DMAs are issued behind the first reads.
    f16x8 af[MX ? 4 : 2][TM], bf[MX ? 4 : 2][TN];           // MX tiles keep all four k-steps' fragments in flight
    auto load_frags = [&](int buf, const char *sb, int ks) {
        const int c = ((2 * ks + lh) ^ fsw) * 16;
#pragma unroll
        for (int t = 0; t < TM; ++t) af[buf][t] = *(const f16x8 *)(sb + a_off[t] + c);
#pragma unroll
        for (int t = 0; t < TN; ++t) bf[buf][t] = *(const f16x8 *)(sb + b_off[t] + c);
    };
    static_assert(NS == 2 || (NS == 3 && (NA + NB == 8 || NA + NB == 6 || NA + NB == 4)), "stage count / DMAs per stage");
    const int mxper = MX && p.mx_period > 0 ? p.mx_period : nk;
    const int n16 = MX && p.nk16 > 0 && p.nk16 < mxper ? p.nk16 : mxper;
    const int mxa = p.mx_scale_a * 0x01010101, mxb = p.mx_scale_b * 0x01010101;
    if (k1 > k0) stage(0, k0);
    if constexpr (NS == 3) { if (k1 - k0 > 1) stage(1, k0 + 1); }
    int cbuf = 0;                                           // buffer of tile kt
    // one K tile (fp16, or - MX builds - 128 e4m3 bytes per row).  The two kinds run in loops of their own below: with one loop and a branch
    // per tile the register allocator carried the accumulators across the back edge in VGPRs and copied all 64 of them into the AGPRs
    // and back EVERY K tile (128 v_accvgpr moves per 16-24 MFMAs in the MX build of the 128 x 128 tile - round 4, found in the ISA).
    auto ktile = [&](auto fp8_tag, int kt) {
        constexpr bool FP8 = decltype(fp8_tag)::value;
        if (NS == 3 && kt + 1 < k1) vm_wait_halftiles((NA + NB) / 2);   // all but the newest stage (NA + NB DMAs) landed
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // NS == 3: a bare barrier - __syncthreads() carries a workgroup fence for which the compiler drains vmcnt, i.e. the stage the
        // counted wait above leaves in flight (round 3: that is why the three-stage builds never beat the two-stage ones)
        if constexpr (NS == 3) PB_BAR_GENERIC();
        else __syncthreads();
        const char *sb = smem + cbuf * STAGE;
        load_frags(0, sb, 0);
        if constexpr (NS == 3) {
            int nb = cbuf + 2; nb = nb >= 3 ? nb - 3 : nb;
            if (kt + 2 < k1) stage(nb, kt + 2);
            cbuf = cbuf == 2 ? 0 : cbuf + 1;
        } else {
            if (kt + 1 < k1) stage(cbuf ^ 1, kt + 1);
            cbuf ^= 1;
        }
        if constexpr (FP8) {
            // MX-fp8 tile (gemm.h nk16): the fragments of k-steps 2q, 2q + 1 are the 32-byte operands of one scaled MFMA
            // the second MFMA group's fragments are read from the LDS under the first group's MFMAs
            constexpr int FB = MX ? 2 : 0;
            load_frags(1, sb, 1);
            load_frags(FB, sb, 2);
            load_frags(FB + 1, sb, 3);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = mfma_mx8(af[0][i], af[1][i], bf[0][j], bf[1][j], acc[i][j], mxa, mxb);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = mfma_mx8(af[FB][i], af[FB + 1][i], bf[FB][j], bf[FB + 1][j], acc[i][j], mxa, mxb);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) load_frags((ks + 1) & 1, sb, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if constexpr (!MX) {
        for (int kt = k0; kt < k1; ++kt) ktile(std::false_type{}, kt);
    } else {
        for (int t0 = 0; t0 < nk; t0 += mxper) {
            const int e16 = t0 + n16 < nk ? t0 + n16 : nk, e8 = t0 + mxper < nk ? t0 + mxper : nk;
            for (int kt = t0 > k0 ? t0 : k0; kt < (e16 < k1 ? e16 : k1); ++kt) ktile(std::false_type{}, kt);
            for (int kt = e16 > k0 ? e16 : k0; kt < (e8 < k1 ? e8 : k1); ++kt) ktile(std::true_type{}, kt);
        }
    }
    __syncthreads();            // staging buffers are dead; reuse LDS for the epilogue patches

    if (p.sk_phase == 1) {
        // split-K, first launch: park this slice's accumulators (lane-major: a wave writes 256 contiguous bytes per register); the second
        // launch adds the slices up in slice order and runs the epilogue (the kernel boundary orders the two across XCDs)
        constexpr int NACC = TM * TN * 16;
        float *mine = p.sk_ws + ((int64_t)(kslice * ntile + swz) * NACC) * NT + tid;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(int64_t)((i * TN + j) * 16 + r) * NT] = acc[i][j][r];
        return;
    }

    if constexpr (EPI == EPI_RESID) resid_io<TM, TN, true>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, lane);
    else run_epilogue<EPI, TM, TN, MX>(p, acc, smem, wave, lane, m0 + wm * TM * 32, n0 + wn * TN * 32, n0);
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 x 64 "ping-pong" kernel: 8 waves = 2 wave groups (rows 0-127 / 128-255) staggered by
// one barrier, so on every SIMD one wave runs its 8-MFMA cluster while its partner issues the
// ds_reads and LDS-DMA of its next cluster.  Per K tile each wave walks its 128 x 64 output in four
// 64 x 32 quadrants (phases p0..p3):
//     p0: read B(j0)[4] + A(i0)[8] | mfma (i0,j0)      p1: read B(j1)[4] | mfma (i0,j1)
//     p2: read A(i1)[8]            | mfma (i1,j1)      p3: -              | mfma (i1,j0)
// LDS: 2 K-tile buffers x {A 32 KB, B 32 KB}; each buffer is staged as four 16 KB "half tiles"
// A_i = rows {64 i .. +64} of both wave groups, B_j = rows {64 wc + 32 j .. +32} of the four wave
// columns, one half tile (2 global_load_lds per thread) per phase, in consumption order and 5-6
// phases ahead of first use:
//     p0: B_1(t+1)   p1: A_1(t+1)   p2: A_0(t+2)   p3: B_0(t+2)
// A region is re-staged >= 2 phases after its last ds_read (WAR) and read >= 1 phase after the
// issuing waves' counted s_waitcnt vmcnt + barrier (RAW); vmcnt never drains to 0 in steady state
// (8 DMAs = 4 half tiles stay in flight).
// ------------------------------------------------------------------------------------------------

#define PB_BAR()                                  \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        asm volatile("s_barrier" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

// VAR != 0 are timing-only ablations (wrong results): 1 no DMA in the loop, 2 no vmcnt waits, 3 no ds_reads,
// 5 DMA + barriers only (no ds_reads, no MFMAs), 6 MFMAs + barriers only.
//
// PERSISTENT WORKGROUPS (round 4).  The grid is one workgroup per CU (launch_g8_impl; PB_GEMM_PERSIST=0 restores one workgroup per
// tile - the loop below then runs once) and a workgroup walks the tiles vb = blockIdx.x + k * gridDim.x of the same XCD-contiguous
// order as before.  What that buys (profiles/r04a_store_probe.txt, r04a_gemm8_phase_cycles.log): a one-tile workgroup cannot end
// before its 128-512 KB of output have left the CU, and with every CU reaching its epilogue at the same moment that is the whole
// chip's write burst at HBM speed - 11-14k of a 54k-cycle fc1 tile - plus 3.4k cycles of prologue DMA latency in front of the next
// workgroup's first MFMA.  A persistent workgroup only has to ISSUE its stores (2-3k cycles) and they drain under the next tile's
// K loop; and the next tile's first six half tiles are requested BEFORE the epilogue (the staging buffers are dead by then), so they
// have landed when the epilogue's last store has been issued:
//   * vmcnt counts loads and stores in issue order and saturates at 63, so once >= PB_EPI_MIN_OPS (56) memory instructions have been
//     issued behind the twelve prefetch DMAs, `s_waitcnt vmcnt(56)` certifies the DMAs without waiting for the newest 56 stores;
//     only epilogues that use no LDS and issue that many stores on every lane take this path (`pf` below): interior tiles of the
//     direct fp16 / fp32-residual epilogues.  Everything else runs the old order: epilogue, barrier, prologue.
//   * K tile 0 of a prefetched tile skips its four counted waits (everything it reads was certified above; the waits would
//     otherwise be the first thing to wait for ALL of the previous tile's stores); the first wait that needs those stores gone is
//     K tile 1's, a whole K tile (~2.4k cycles) after the epilogue.
template <int AMODE, int EPI, int VAR = 0, bool BUFP = false, bool MX = false>
__global__ __launch_bounds__(512) void gemm8_kernel(const GemmArgs p) {
    constexpr int BM = 256, BN = 256;
    constexpr int BUF = 65536, BOFF = 32768;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wr = wave >> 2, wc = wave & 3;
    long long ts0 = 0, ts1 = 0, ts2 = 0, tr0 = 0;
    stagger_start(p.stagger, 256);
    if (p.dbg) { ts0 = __builtin_readcyclecounter(); tr0 = wall_clock64(); }

    const int tilesN = (p.N + BN - 1) / BN;
    const int nwg = p.ntiles > 0 ? p.ntiles : (int)gridDim.x;     // tiles of the launch (the grid may be smaller: persistent workgroups)
    const int qd = nwg >> 3, rm = nwg & 7;
    int m0 = 0, n0 = 0, swz = 0;

    // ---- staging geometry: half tile h in {A_0, A_1, B_0, B_1}, two DMAs (u = 0, 1) per thread ----
    // DMA (wave, u) covers the 8 LDS rows starting at row0; lane -> row0 + (lane >> 3), chunk lane & 7.
    const int cld = p.cLd ? p.cLd : p.cC, padx = p.cPadX >= 0 ? p.cPadX : p.cPad;
    int a_row0[2][2], b_row0[2][2];                  // [half][u], tile-local row of the DMA's first row
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int g = wave * 2 + u;
            a_row0[hf][u] = (g < 8 ? 0 : 128) + hf * 64 + (g & 7) * 8;
            b_row0[hf][u] = (g >> 2) * 64 + hf * 32 + (g & 3) * 8;
        }
    const f16 *a_ptr[2][2];
    // conv: which taps of the output pixel lie inside the image (tap_mask; 0 for rows >= M) - one register per DMA, tested against the
    // K tile's two tap bits with an AND and a compare
    int a_yx[2][2];
    const f16 *b_ptr[2][2];
    const int nk = p.K >> 6;
    // BUFP: stage through the buffer path (`buffer_load_dwordx4 ... lds`) - measurably cheaper to issue than the flat
    // `global_load_lds` (8192^3: 1145 -> 1245 TF, qkv / fc1 shapes +14 ... +17 %).  The resource covers the whole operand
    // (dense, weights) or the one or two images this tile's rows fall into (conv: a whole DPT map batch exceeds 4 GB);
    // the launcher only picks this variant when those spans fit 32-bit byte offsets.  Out-of-range offsets read zeros,
    // which is how padded taps and rows >= M are fed.
    __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, 0u), rsW = make_rsrc(p.W, 0u);
    unsigned a_voff[2][2], b_voff[2][2];
    if constexpr (BUFP) {
        rsW = make_rsrc(p.W, (unsigned)((int64_t)((p.N + 255) / 256 * 256) * p.K * 2));
        if constexpr (AMODE == A_DENSE) rsA = make_rsrc(p.A, (unsigned)((int64_t)p.M * p.lda * 2));
    }
    // staging past the last K tile re-reads the last one into a slot nobody reads any more: the loop stays branch free
    // and every phase can use the same counted wait
    // conv: each A half has its own tap cursor (ky, kx, c0) that steps one K tile per call - no divisions in the loop - and
    // the per-lane pixel offset of tap (0, 0) is precomputed, so a DMA costs one add, two range tests and a select
    // (round 4: the walk is a table in the LDS, conv_ktab_entry; A_0(k) fetches K tile k's word, A_1(k) - always the next stage_a call - re-uses it)
    const unsigned *ktab = (const unsigned *)(smem + 2 * BUF);
    unsigned e_nxt = 0, e_cur = 0;
    int a_pix0[2][2];

    // tile vb of the launch -> (m0, n0) and this thread's staging addresses.  XCD-contiguous remap (bijective for any tile count;
    // workgroup b runs on XCD b % 8 and the persistent stride is a multiple of 8), then, within an XCD's contiguous range, a walk in
    // 8 (M) x GN (N) super-tiles so that the ~32 tiles in flight on the XCD share few A and W panels (fewer L2 misses -> less
    // MALL/HBM traffic; loop time is unchanged).
    auto setup = [&](const GemmArgs &p, int vb) {
        // everything lane dependent is derived from an opaque copy of the lane id, i.e. recomputed per tile instead of being kept in
        // a dozen VGPRs across the K loops
        int sl = lane;
        asm volatile("" : "+v"(sl));
        const int lrow = sl >> 3;
        // swizzled global chunk of this lane's LDS slot: row0 is a multiple of 8 with (row0 >> 3) & 1 == u
        int cgu[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) cgu[u] = (sl & 7) ^ ((4 * u + (sl >> 4)) & 7);
        const int xcd = vb & 7;
        swz = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (vb >> 3);
        int tile_m, tile_n;
        {
            const int GN = tilesN < 4 ? tilesN : 4;
            const int per_band = 8 * tilesN;                 // tiles in a band of 8 M-panels
            const int band = swz / per_band, rem = swz - band * per_band;
            const int tilesM = nwg / tilesN;
            const int bh = (tilesM - band * 8) < 8 ? (tilesM - band * 8) : 8;   // M-panels in this band
            const int grp = rem / (bh * GN), r2 = rem - grp * bh * GN;
            const int gw = (tilesN - grp * GN) < GN ? (tilesN - grp * GN) : GN; // N-panels in this group
            tile_m = band * 8 + r2 / gw;
            tile_n = grp * GN + r2 % gw;
        }
        m0 = tile_m * BM; n0 = tile_n * BN;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int m = m0 + a_row0[hf][u] + lrow;
                if constexpr (AMODE == A_DENSE) {
                    const int mc = m < p.M ? m : p.M - 1;
                    a_ptr[hf][u] = p.A + (int64_t)mc * p.lda + cgu[u] * 8;
                    a_yx[hf][u] = a_pix0[hf][u] = 0;
                } else {
                    const int ohw = p.cOH * p.cOW;
                    const int b = m / ohw, rem = m - b * ohw;
                    const int oy = rem / p.cOW, ox = rem - oy * p.cOW;
                    a_ptr[hf][u] = p.A + (int64_t)b * p.cH * p.cW * cld + cgu[u] * 8;
                    const int iy0 = oy * p.cStride - p.cPad, ix0 = ox * p.cStride - padx;
                    a_yx[hf][u] = (int)(m < p.M ? tap_mask(iy0, ix0, p.cH, p.cW) : 0u);
                    a_pix0[hf][u] = (iy0 * p.cW + ix0) * cld;
                }
                b_ptr[hf][u] = p.W + (int64_t)(n0 + col_map(b_row0[hf][u] + lrow, epi_interleaved<EPI, 2>())) * p.K + cgu[u] * 8;
            }
        if constexpr (BUFP) {
            if constexpr (AMODE == A_CONV) {
                const int ohw = p.cOH * p.cOW, nimg = p.M / ohw, b0 = p.bufmode == 2 ? m0 / ohw : 0;
                const int64_t img = (int64_t)p.cH * p.cW * cld;
                rsA = make_rsrc(p.A + b0 * img, (unsigned)((p.bufmode == 2 ? (nimg - b0 < 2 ? nimg - b0 : 2) : nimg) * img * 2));
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if constexpr (AMODE == A_DENSE) {
                        a_voff[hf][u] = (unsigned)((a_ptr[hf][u] - p.A) * 2);
                    } else {
                        // byte offset of tap (0, 0) of this lane's pixel inside the resource (the pixel part may be "negative": only
                        // in-range taps use the sum, and that sum is in range)
                        const int ohw = p.cOH * p.cOW, b0 = p.bufmode == 2 ? m0 / ohw : 0;
                        const int m = m0 + a_row0[hf][u] + lrow;
                        a_voff[hf][u] = (unsigned)((int64_t)(m / ohw - b0) * p.cH * p.cW * cld * 2) + cgu[u] * 16 + (unsigned)(a_pix0[hf][u] * 2);
                    }
                    b_voff[hf][u] = (unsigned)((b_ptr[hf][u] - p.W) * 2);
                }
        }
    };
    auto stage_a = [&](int hf, int kt_) {
        const int ktc = kt_ < nk ? kt_ : nk - 1;
        const int kt = (p.kwrap && ktc >= p.kwrap) ? ktc - p.kwrap : ktc;         // split-fp16 segments re-read A (gemm.h)
        char *base = smem + (kt_ & 1) * BUF;
        unsigned tapoff2 = 0, tapsel = 0;
        if constexpr (AMODE == A_CONV) {
            if (hf == 0) {
                e_cur = (unsigned)__builtin_amdgcn_readfirstlane((int)e_nxt);
                e_nxt = ktab[ktc + 1 < nk ? ktc + 1 : nk - 1];
            }
            tapoff2 = ktab_bytes(e_cur); tapsel = ktab_sel(e_cur);
#ifdef PB_DIAG
            // PB_GEMM_ABL=4, see gemm_kernel's stage() (modes 8 and 16 cost this kernel registers it does not have: spills inside the K loop)
            tapsel = ((p.ablate & 4) && (tapsel >> 8) != 1u) ? 0xFFFFFFFFu : tapsel;
#endif
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int g = wave * 2 + u;
            char *dst = base + ((g < 8 ? 0 : 128) + hf * 64 + (g & 7) * 8) * 128;
            if constexpr (AMODE == A_DENSE) {
                if constexpr (BUFP) glds16_buf(rsA, (int)a_voff[hf][u], kt * 128, dst);
                else glds16(a_ptr[hf][u] + kt * 64, dst);
            } else {
                const bool ok = ((unsigned)a_yx[hf][u] & tapsel) == tapsel;            // tap_mask: an AND and a compare per DMA
                if constexpr (BUFP) {
                    glds16_buf(rsA, (int)(ok ? a_voff[hf][u] + tapoff2 : 0xFFFFFF00u), 0, dst);      // any out-of-range offset reads zeros
                } else {
                    glds16(ok ? a_ptr[hf][u] + (a_pix0[hf][u] + (int)(tapoff2 >> 1)) : p.zero, dst);
                }
            }
        }
    };
    auto stage_b = [&](int hf, int kt_) {
        const int kt = kt_ < nk ? kt_ : nk - 1;
        char *base = smem + (kt_ & 1) * BUF + BOFF;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int g = wave * 2 + u;
            if constexpr (BUFP) glds16_buf(rsW, b_voff[hf][u], kt * 128, base + ((g >> 2) * 64 + hf * 32 + (g & 3) * 8) * 128);
            else glds16(b_ptr[hf][u] + kt * 64, base + ((g >> 2) * 64 + hf * 32 + (g & 3) * 8) * 128);
        }
    };
    // a tile's first six half tiles: A_0(0) B_0(0) B_1(0) A_1(0) A_0(1) B_0(1)   (issuing these BEFORE the residual loads measured 5 % slower on proj)
    auto prologue = [&]() {
        if constexpr (AMODE == A_CONV) e_nxt = ktab[0];
        stage_a(0, 0); stage_b(0, 0); stage_b(1, 0); stage_a(1, 0);
        stage_a(0, 1); stage_b(0, 1);
    };

    // ---- fragment addressing: chunk(ks) = (lh ^ fsw) ^ 2 ks  ->  byte offset = c0 ^ (32 ks) ----
    const int li = lane & 31, lh = lane >> 5;
    const int c0 = (lh ^ ((li >> 1) & 7)) * 16;
    const int a_base = (wr * 128 + li) * 128;            // + i*8192 + rt*4096
    const int b_base = BOFF + (wc * 64 + li) * 128;      // + j*4096

    f32x16 acc[4][2];
    f16x8 fa[2][4], fb0[4], fb1[4];

    const int mxa = p.mx_scale_a * 0x01010101, mxb = p.mx_scale_b * 0x01010101;
    // one K tile; FP8 tiles hold 128 e4m3 bytes per row and go through the MX-scaled MFMA (gemm.h nk16): same staging, same fragment
    // reads, 4 MFMAs of 64 cycles instead of 8 of 32 per phase - twice the K per tile at the same matrix-pipe time.
    // FIRST (K tile 0) with `nw` set: the tile's operands were prefetched and certified before the loop (see the header), its four
    // counted waits are skipped.
    auto tile = [&](auto fp8_tag, auto first_tag, int t, bool nw) {
        constexpr bool FP8 = decltype(fp8_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        // K tile 0 starts every accumulator from the inline constant 0 (no 128-register clear per tile); the residual epilogue starts
        // them from the residual tile instead.  VAR != 0 ablations skip MFMAs, so they keep the explicit clear.
        constexpr bool ZC = FIRST && EPI != EPI_RESID && VAR == 0;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const char *sb = smem + (t & 1) * BUF;
        // ================= p0 =================
        if ((VAR != 3 && VAR != 5 && VAR != 6) || t == 0) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb0[ks] = *(const f16x8 *)(sb + b_base + (c0 ^ (ks * 32)));
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[rt][ks] = *(const f16x8 *)(sb + a_base + rt * 4096 + (c0 ^ (ks * 32)));
        }
        if (VAR != 1 && VAR != 6) stage_b(1, t + 1);
        if (VAR == 0 || VAR == 3 || VAR == 5) { if (!(FIRST && nw)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        PB_BAR();
        __builtin_amdgcn_s_setprio(1);
        if (VAR != 5) {
            if constexpr (FP8) {
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc[rt][0] = mfma_mx8(fa[rt][2 * q2], fa[rt][2 * q2 + 1], fb0[2 * q2], fb0[2 * q2 + 1], acc[rt][0], mxa, mxb);
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[rt][ks], fb0[ks], ZC && ks == 0 ? zero16 : acc[rt][0], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        PB_BAR();
        // ================= p1 =================
        if ((VAR != 3 && VAR != 5 && VAR != 6) || t == 0) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb1[ks] = *(const f16x8 *)(sb + b_base + 4096 + (c0 ^ (ks * 32)));
        }
        if (VAR != 1 && VAR != 6) stage_a(1, t + 1);
        if (VAR == 0 || VAR == 3 || VAR == 5) { if (!(FIRST && nw)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        PB_BAR();
        __builtin_amdgcn_s_setprio(1);
        if (VAR != 5) {
            if constexpr (FP8) {
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc[rt][1] = mfma_mx8(fa[rt][2 * q2], fa[rt][2 * q2 + 1], fb1[2 * q2], fb1[2 * q2 + 1], acc[rt][1], mxa, mxb);
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[rt][ks], fb1[ks], ZC && ks == 0 ? zero16 : acc[rt][1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        PB_BAR();
        // ================= p2 =================
        if (VAR != 3 && VAR != 5 && VAR != 6)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                fa[rt][ks] = *(const f16x8 *)(sb + a_base + 8192 + rt * 4096 + (c0 ^ (ks * 32)));
        if (VAR != 1 && VAR != 6) stage_a(0, t + 2);
        if (VAR == 0 || VAR == 3 || VAR == 5) { if (!(FIRST && nw)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        PB_BAR();
        __builtin_amdgcn_s_setprio(1);
        if (VAR != 5) {
            if constexpr (FP8) {
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc[2 + rt][1] = mfma_mx8(fa[rt][2 * q2], fa[rt][2 * q2 + 1], fb1[2 * q2], fb1[2 * q2 + 1], acc[2 + rt][1], mxa, mxb);
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc[2 + rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[rt][ks], fb1[ks], ZC && ks == 0 ? zero16 : acc[2 + rt][1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        PB_BAR();
        // ================= p3 =================
        if (VAR != 1 && VAR != 6) stage_b(0, t + 2);
        if (VAR == 0 || VAR == 3 || VAR == 5) { if (!(FIRST && nw)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        PB_BAR();
        __builtin_amdgcn_s_setprio(1);
        if (VAR != 5) {
            if constexpr (FP8) {
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc[2 + rt][0] = mfma_mx8(fa[rt][2 * q2], fa[rt][2 * q2 + 1], fb0[2 * q2], fb0[2 * q2 + 1], acc[2 + rt][0], mxa, mxb);
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        acc[2 + rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[rt][ks], fb0[ks], ZC && ks == 0 ? zero16 : acc[2 + rt][0], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        PB_BAR();
    };

    if constexpr (AMODE == A_CONV) {                     // the K walk of the launch (conv_ktab_entry), once per workgroup
        for (int t = tid; t < nk; t += 512) ((unsigned *)(smem + 2 * BUF))[t] = conv_ktab_entry(p, cld, t);
        __syncthreads();                                 // (moving this barrier behind the tile set-up measured slower: start -> loop 9.9k -> 11.7k cycles, r04w)
    }
    int vb = blockIdx.x;
    setup(p, vb);
    bool pf = false;                                     // this tile's prologue was issued before the previous tile's epilogue
    bool swapped = false;                                // EPI_RESID: this tile's residual rows were loaded by the previous tile's epilogue
    if constexpr (EPI != EPI_RESID) prologue();
    while (true) {
        if constexpr (EPI != EPI_RESID && VAR != 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        if constexpr (EPI == EPI_RESID) {
            if (!swapped) {                                  // (after resid_swap the accumulators already hold this tile's residual rows)
                if (p.resid_atomic) resid_io_atomic<4, 2, false>(p, acc, m0 + wr * 128, n0 + wc * 64, lane);
                else resid_io<4, 2, false>(p, acc, m0 + wr * 128, n0 + wc * 64, lane);
            }
            if (!pf) prologue();
        }
        if (pf) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");   // >= 56 stores were issued behind the twelve DMAs: they have landed
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // A_0(0), B_0(0) landed (this wave's share)
        PB_BAR();
        if (wr == 1) PB_BAR();                               // stagger the second wave group by one barrier
        if (p.dbg) ts1 = __builtin_readcyclecounter();

        // K tile 0 is an fp16 tile in every layout (gemm.h nk16 >= 1 when MX tiles exist)
        tile(std::false_type{}, std::true_type{}, 0, pf);
        if constexpr (!MX) {
            for (int t = 1; t < nk; ++t) tile(std::false_type{}, std::false_type{}, t, false);
        } else if constexpr (AMODE == A_DENSE) {     // [fp16 tiles | fp8 tiles] once
            const int n16 = p.nk16 > 0 && p.nk16 < nk ? p.nk16 : nk;
            for (int t = 1; t < n16; ++t) tile(std::false_type{}, std::false_type{}, t, false);
            if constexpr (VAR == 0) {
                for (int t = n16; t < nk; ++t) tile(std::true_type{}, std::false_type{}, t, false);
            }
        } else {                                     // per tap (mx_period tiles): fp16 tiles, then fp8 tiles
            const int per = p.mx_period > 0 ? p.mx_period : nk;
            const int n16 = p.nk16 > 0 && p.nk16 < per ? p.nk16 : per;
            for (int t0 = 0; t0 < nk; t0 += per) {
                for (int t = t0 ? t0 : 1; t < t0 + n16; ++t) tile(std::false_type{}, std::false_type{}, t, false);
                for (int t = t0 + n16; t < t0 + per; ++t) tile(std::true_type{}, std::false_type{}, t, false);
            }
        }
        if (wr == 0) PB_BAR();                               // re-align the two wave groups
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (p.dbg) ts2 = __builtin_readcyclecounter();

        // ---- this tile's epilogue, with the next tile's first DMAs in front of it where the epilogue allows ----
        const int em0 = m0, en0 = n0, eswz = swz;
        vb += gridDim.x;
        const bool more = vb < nwg;
        // epilogues that touch no LDS and issue >= PB_EPI_MIN_OPS stores on every lane: interior tiles of the direct fp16 epilogues
        // (64+ stores; the V third of q/k/v transposes through LDS, a pixel-shuffle with ps_co % 64 != 0 takes the LDS path) and of
        // the fp32 residual epilogue (128 stores)
        bool lds_free = EPI == EPI_RESID || EPI == EPI_STD || EPI == EPI_QKV || EPI == EPI_PIXSHUF;
        if constexpr (EPI == EPI_QKV) lds_free = en0 < 2 * p.D;
        if constexpr (EPI == EPI_PIXSHUF) lds_free = (p.ps_co & 63) == 0;
        pf = more && p.prefetch && lds_free && em0 + BM <= p.M && en0 + BN <= p.N;
        // the epilogue and the tile set-up read the kernel arguments through an opaque copy of the kernarg segment pointer: their
        // fields are s_load'ed per tile instead of being hoisted out of the tile loop, where ~100 of them would have to stay in
        // SGPRs across the K loop (measured: 95-127 SGPRs spilled into VGPR lanes, 12-38 VGPRs spilled behind them)
        typedef const __attribute__((address_space(4))) GemmArgs *kargs_t;
        kargs_t ke = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ke));
        // ... and through a by-value COPY of the struct (only the fields that are used get loaded, once): read through the pointer, every
        // field would be re-loaded after every store of the epilogue (the stores may alias it as far as the compiler knows) - measured as
        // 7.0k instead of 2.4k cycles for the 64 stores of a plain fp16 tile (profiles/r04d_gemm8_phase_cycles.log)
        struct ArgWords { int w[sizeof(GemmArgs) / 4]; };
        static_assert(sizeof(GemmArgs) % 4 == 0 && sizeof(ArgWords) == sizeof(GemmArgs), "GemmArgs is copied word by word");
        ArgWords aw;
#pragma unroll
        for (int i = 0; i < (int)(sizeof(GemmArgs) / 4); ++i) aw.w[i] = ((const __attribute__((address_space(4))) int *)ke)[i];
        const GemmArgs pe = __builtin_bit_cast(GemmArgs, aw);
        if (pf) { setup(pe, vb); prologue(); }
        // the epilogue sees an opaque copy of the lane id: nothing it derives from it is loop invariant, so the compiler does not hoist
        // epilogue address arithmetic out of the tile loop and carry it (in VGPRs the K loop needs) across the MFMA phases
        int elane = lane;
        asm volatile("" : "+v"(elane));
#ifdef PB_DIAG
        if (!MX && (pe.ablate & 3)) {
#else
        if (!MX && pe.ablate) {                  // (fp16-only builds: the MX builds have no register to spare for a diagnostic)
#endif
            // timing-only ablations (PB_GEMM_ABL): what a tile costs with no epilogue at all (1) / with the cheapest imaginable one (2:
            // 64 packed-fp16 buffer stores in the MFMA layout, no bias, no activation, no copies)
            if (pe.ablate == 2 && pe.out) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(pe.out, (unsigned)((int64_t)pe.M * pe.ldo * 2));
                const int ldo = (int)pe.ldo;
                const unsigned voff = (unsigned)(((em0 + wr * 128 + 4 * (elane >> 5)) * ldo + en0 + wc * 64 + 2 * (elane & 31)) * 2);
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        f16x2 o; o[0] = (f16)acc[tm][0][r]; o[1] = (f16)acc[tm][1][r];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rs, (int)voff, (tm * 32 + (r & 3) + 8 * (r >> 2)) * ldo * 2, 0);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
            }
        } else if constexpr (EPI == EPI_RESID) {
            // pf implies that this tile is interior and that setup() already moved (m0, n0) to the next tile
            swapped = pf && !pe.resid_atomic && m0 + BM <= pe.M && n0 + BN <= pe.N;
            if (swapped) resid_swap<4>(pe, acc, em0 + wr * 128, en0 + wc * 64, m0 + wr * 128, n0 + wc * 64, elane);
            else if (pe.resid_atomic) resid_io_atomic<4, 2, true>(pe, acc, em0 + wr * 128, en0 + wc * 64, elane);
            else resid_io<4, 2, true>(pe, acc, em0 + wr * 128, en0 + wc * 64, elane);
        }
        else run_epilogue<EPI, 4, 2, MX>(pe, acc, smem, wave, elane, em0 + wr * 128, en0 + wc * 64, en0);
        if (p.dbg && tid == 0) {
            // per TILE: start (kernel start or the previous tile's epilogue issued), loop start, loop end, epilogue issued (twice: a
            // persistent workgroup never waits for its stores), wall clock at start / end, tile index
            long long *d = p.dbg + (long long)eswz * 8;
            const long long t_issue = __builtin_readcyclecounter();
            d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = t_issue; d[4] = tr0; d[5] = wall_clock64();
            d[6] = t_issue; d[7] = eswz;
            ts0 = t_issue; tr0 = d[5];
        }
        if (!more) break;
        // the next tile's staging addresses are (re)computed HERE rather than carried across the epilogue: with them live the epilogue
        // spills (the kernel sits at 237-256 VGPRs); the opaque copy of vb keeps the compiler from re-using the values of the call above
        asm volatile("" : "+s"(vb));
        setup(pe, vb);
        if (!pf) {
            __syncthreads();                                 // the LDS patches of this tile's epilogue are dead
            if constexpr (EPI != EPI_RESID) prologue();
        }
    }
}

template <int AMODE, int EPI, bool MX, bool BUFP>
int launch_g8_impl(hipStream_t stream, const GemmArgs &a) {
    constexpr int SMEM = 131072 + (AMODE == A_CONV ? KTAB_BYTES : 0);
    auto kern = gemm8_kernel<AMODE, EPI, 0, BUFP, MX>;
    static char name[96];
    if (!name[0]) snprintf(name, sizeof(name), "gemm8_kernel<%d, %d, 0, %s, %s>", AMODE, EPI, BUFP ? "true" : "false", MX ? "true" : "false");
    pb_gemm_set_last_kernel(name);
    static bool attr_set = false;
    if (!attr_set) {
        PB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set = true;
    }
    const int tilesM = (a.M + 255) / 256, tilesN = (a.N + 255) / 256;
    // persistent workgroups: one per CU walks the tiles (gemm8_kernel header); PB_GEMM_PERSIST=0 launches one workgroup per tile,
    // PB_GEMM_PREFETCH=0 keeps the persistent loop but issues every tile's prologue after the previous epilogue (A/B switches)
    static int ncu = 0, persist = 0, prefetch = 1;
    if (!ncu) {
        int dev = 0;
        PB_HIP(hipGetDevice(&dev));
        PB_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        // default ON since the end of round 4.  History: on the first straight-line epilogues the per-tile stamps had a one-tile workgroup as
        // fast or faster (profiles/r04f_gemm8_phase_cycles.log: fc1 45.1k against 46.9k cycles per tile, proj 77.7k against 81.6k - what a
        // persistent workgroup saves in front of the K loop it pays inside the epilogue, whose own loads wait behind the prefetched DMAs on the
        // one in-order vmcnt counter) and the bench saw no difference; on the final tree the bench does (r04w, alternating on one box:
        // 120.55 -> 121.17 frames/s; proj + fc2 28.2 -> 27.6 ms, the update block's convolutions 35.5 -> 34.9, fc1 19.5 -> 19.4) although a
        // tile's own stamps are still ~3 % longer: launch ramp and tail, which the stamps do not see, shrink.  PB_GEMM_PERSIST=0: one workgroup per tile.
        persist = pb_env_int("PB_GEMM_PERSIST", 1);
        prefetch = pb_env_int("PB_GEMM_PREFETCH", 1);
    }
    GemmArgs b = a;
    b.ntiles = tilesM * tilesN;
    b.prefetch = prefetch;
    // b.ablate: timing-only switch of the microbenchmark op (pb_op_gemm_bench reads PB_GEMM_ABL); engine launches leave it 0
    static int ratomic = -1;
    if (ratomic < 0) ratomic = pb_env_int("PB_RESID_ATOMIC", 0);
    b.resid_atomic = EPI == EPI_RESID ? ratomic : 0;
    const int grid = persist && b.ntiles > ncu ? ncu : b.ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SMEM, stream, b);
    PB_HIP(hipGetLastError());
    return 0;
}

// The buffer-path variants need every DMA offset to fit an unsigned 32-bit byte count: 1 = the whole A operand does,
// 2 = (conv) any two consecutive images do and a BM-row tile never spans more than two, 0 = neither (flat path).
inline int buffer_mode(int amode, const GemmArgs &a, int BM) {
    static int env = -1;
    if (env < 0) { const char *e = getenv("PB_GEMM_BUFFER"); env = e ? atoi(e) : 1; }
    if (!env) return 0;
    const int64_t lim = (1LL << 32) - (1 << 20);
    if ((int64_t)((a.N + 255) / 256 * 256) * a.K * 2 >= lim) return 0;
    if (amode == A_DENSE) return (int64_t)a.M * a.lda * 2 < lim ? 1 : 0;
    const int cld = a.cLd ? a.cLd : a.cC;
    const int64_t ohw = (int64_t)a.cOH * a.cOW, img = (int64_t)a.cH * a.cW * cld * 2;
    if (ohw <= 0 || a.M % ohw != 0) return 0;
    if ((a.M / ohw) * img < lim) return 1;
    return ohw >= BM && 2 * img < lim ? 2 : 0;
}

template <int AMODE, int EPI, bool MX>
int launch_g8(hipStream_t stream, const GemmArgs &a) {
    GemmArgs b = a;
    b.bufmode = buffer_mode(AMODE, a, 256);
    if (b.bufmode) return launch_g8_impl<AMODE, EPI, MX, true>(stream, b);
    return launch_g8_impl<AMODE, EPI, MX, false>(stream, a);
}

template <int BM, int BN, int WM, int WN, int AMODE, int EPI, bool MX, bool BUFP = false, int NS = 2>
int launch_t(hipStream_t stream, const GemmArgs &a) {
    constexpr int NT = WM * WN * 64;
    constexpr int TN = BN / WN / 32;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int EPIB = (EPI == EPI_QKV) ? TN * 32 * 36 * 4 : 32 * (TN * 32 + 4) * 4;
    constexpr int STG = NS * STAGE + (AMODE == A_CONV && BN != 64 ? (BN == 96 ? KTAB_CW_BYTES : KTAB_BYTES) : 0);          // (+ the conv K-walk table, gemm_kernel KT)
    constexpr int SMEM = STG > WM * WN * EPIB ? STG : WM * WN * EPIB;
    if constexpr (!BUFP && ((BM == 128 && BN == 128) || NS == 3 || BN == 64 || BN == 96)) {       // the small tiles also have a buffer-path build (BN == 64: 256 x 64 and 64 x 64)
        GemmArgs b = a;
        b.bufmode = buffer_mode(AMODE, a, BM);
        if (b.bufmode) return launch_t<BM, BN, WM, WN, AMODE, EPI, MX, true, NS>(stream, b);
    }
    auto kern = gemm_kernel<BM, BN, WM, WN, AMODE, EPI, BUFP, NS, MX>;
    static char name[112];
    if (!name[0])
        snprintf(name, sizeof(name), "gemm_kernel<%d, %d, %d, %d, %d, %d, %s, %d, %s>", BM, BN, WM, WN, AMODE, EPI, BUFP ? "true" : "false", NS,
                 MX ? "true" : "false");
    pb_gemm_set_last_kernel(name);
    static bool attr_set = false;
    if (!attr_set) {
        PB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set = true;
    }
    const int tilesM = (a.M + BM - 1) / BM, tilesN = (a.N + BN - 1) / BN;
    // split-K launches (gemm.h splitk; launch_gemm sets it for the 128 x 128 tile only) carry `splitk` workgroups per tile
    if (BM == 128 && BN == 128 && NS == 2 && a.splitk > 1) {
        GemmArgs b = a;
        b.sk_phase = 1;
        hipLaunchKernelGGL(kern, dim3(tilesM * tilesN * a.splitk), dim3(NT), SMEM, stream, b);
        b.sk_phase = 2;
        hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(NT), SMEM, stream, b);
        PB_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(NT), SMEM, stream, a);
    PB_HIP(hipGetLastError());
    return 0;
}

template <int AMODE, int EPI, bool MX>
int launch_tile(hipStream_t s, int tile, const GemmArgs &a) {
    if constexpr (EPI == EPI_HEAD) {
        return launch_t<256, 32, 4, 1, AMODE, EPI, MX>(s, a);
    } else {
        if (tile == TILE_256) return launch_g8<AMODE, EPI, MX>(s, a);
        if constexpr (EPI == EPI_STD) {
            if (tile == TILE_256x64) return launch_t<256, 64, 4, 1, AMODE, EPI, MX>(s, a);
            // 64 < N <= 96 (RAFT / GMFlow encoder stage 2): four waves of 32 x 96, a quarter fewer MFMAs and B rows than the 128-wide tile spends on padding columns
            if constexpr (AMODE == A_CONV) if (tile == TILE_128x96) return launch_t<128, 96, 4, 1, AMODE, EPI, MX>(s, a);
            // (round 3, measured and removed: launch_t<256, 128, 4, 2, AMODE, EPI, MX, false, 3> - 8 waves of 64 x 64, three LDS stages two K
            // tiles ahead behind bare barriers, 144 KB, one workgroup per CU - against this tile's two workgroups per CU: update block 35.4 ->
            // 36.7 ms, RAFT encoders 19.1 -> 20.0, DPT head 12.5 -> 13.4 on one box; these launches are not waiting on DMA latency)
        }
        // (round 5, measured and removed: one-wave 64 x 64 workgroups for the residual GEMMs of ONE frame - 624 instead of 160 workgroups on 256 CUs:
        // proj + fc2 2.79 -> 2.42 ms of a 9.0 ms call (profiles/r05h_latency_batch1.txt); the staging swizzle below also assumes NT / 8 rows per
        // DMA step is a multiple of 16, which a 64-thread block breaks)
        return launch_t<128, 128, 2, 2, AMODE, EPI, MX>(s, a);
    }
}

}  // namespace

