// Single-head attention over 128-wide heads: O[b][q] = softmax_k(Q[b][q] . K[b][k] / sqrt(128) + mask(q, k)) V[b][k] - the
// attention of the reference's GMFlow transformer (bands/gmflow/transformer.py:8-15 full, :47-101 split windows with the shifted-window
// mask of :18-44) and, with V = coordinates / flow, of its global matching and flow propagation (matching.py:7-42, transformer.py:316-337).
//
// Flash style, in the layouts tools/mfma_layout.py models and tests/test_mfma_layout_cpu.py runs in numpy:
//   * a wave owns 32 queries (Q fragments of the 8 k-steps in registers), a workgroup 4 waves = 128 queries of one batch element;
//   * per tile of 32 keys, S^T[key][q] = K Q^T with the K rows in swap_bits23 order: a lane holds 16 scores of ONE query column (row
//     max / sum in-lane plus one exchange with lane ^ 32) and registers 8 s .. 8 s + 7 are the B fragment of k-step s of
//     O^T[d][q] += Vt[d][key] P^T[key][q] (NVB 32-row blocks of d: 4 for a 128-wide V, 1 for the 2-column coordinate / flow V);
//   * K and Vt tiles go through the LDS (plain loads + ds_write; rows padded by 16 bytes: conflict-free 16-byte fragment reads),
//     double buffered, one barrier per tile;
//   * scores and the running max live in log2 units (Q . K scaled by log2 e / sqrt(128)); statistics, O and the output are fp32.
// SPLIT (the flow_gmflow band's precision mode, DESIGN.md section 7: this network's two softmax stages amplify operand rounding): Q, K and
// V arrive as hi + lo fp16 pairs and P is split in registers, S = q_hi k_hi + q_lo k_hi + q_hi k_lo and O += v_hi p_hi + v_hi p_lo +
// v_lo p_hi in the same fp32 accumulators - three MFMA passes each, operands good to ~22 bits.
// Not tuned: no LDS-DMA, no matrix-pipe softmax tricks (attention.hip has those for the 64-wide ViT heads).
#include "common.h"
#include "kernels.h"
#include "../../include/prisma_bands.h"

namespace {

constexpr int AQ = 128, AKT = 32, AD = 128;                     // queries per workgroup, keys per tile, head dim

// SPLIT: Q, K as hi + lo (three MFMA passes for S); SPV: V and P as hi + lo (three passes for P V).  The flow_gmflow band runs its window
// attention with SPLIT only - rounding P and V to fp16 there moves the flow by 2.6e-4 of its range (tools/precision_budget_gmflow.py),
// q / k by 5.8e-4 - and the global matching / propagation, whose V are coordinates and flows, with both.
template <bool SPLIT, bool SPV, int NVB>
struct AttnGeom {
    static constexpr int KROW = (SPLIT ? 2 : 1) * AD * 2 + 16;  // LDS row strides in bytes (272 / 528, 80)
    static constexpr int VROW = AKT * 2 + 16;
    static constexpr int VR = (SPV ? 2 : 1) * NVB * 32;         // Vt rows per tile: [hi rows | lo rows]
    static constexpr int K_BYTES = AKT * KROW, V_BYTES = VR * VROW, BUF = K_BYTES + V_BYTES;
};

// Q, K: rows of ldq halfs, the hi part in columns [0, 128), with SPLIT the lo part in [128, 256); batch element b reads queries at
//       Q + b q_bstride and keys / values at index b ^ kxor (K + (b ^ kxor) k_bstride: cross attention between the two frames of a pair).
// Vt:   [.., VR, ldv] fp16 (v_bstride = 0 shares one V between all batch elements): rows [0, NVB * 32) hi, then lo.
// region: [nreg, L] int8 or null; batch element b uses row b % nreg.
// O:    [B, L, NVB * 32] fp32.
template <bool SPLIT, bool SPV, int NVB>
__global__ __launch_bounds__(256) void attn128_kernel(const f16 *__restrict__ Q, const f16 *__restrict__ K, const f16 *__restrict__ Vt,
                                                      const int8_t *__restrict__ region, float *__restrict__ O, int L, int ldq, int ldv,
                                                      int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int kxor, int nreg) {
    using G = AttnGeom<SPLIT, SPV, NVB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * AQ + wave * 32;
    const f16 *Qb = Q + (int64_t)b * q_bstride, *Kb = K + (int64_t)(b ^ kxor) * k_bstride, *Vb = Vt + (int64_t)(b ^ kxor) * v_bstride;
    const int8_t *rg = region ? region + (int64_t)(b % nreg) * L : nullptr;
    const int kperm = (li & 19) | ((li & 4) << 1) | ((li & 8) >> 1);
    constexpr int NP = SPLIT ? 2 : 1, NPV = SPV ? 2 : 1;

    // Q fragments (B operand of S^T): query q0 + li, k-steps 0..7 of the hi (and lo) part
    f16x8 qf[NP][8];
    {
        const int q = q0 + li < L ? q0 + li : L - 1;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qf[p][ks] = *(const f16x8 *)(Qb + (int64_t)q * ldq + p * AD + ks * 16 + lh * 8);
    }
    const int my_region = rg ? rg[q0 + li < L ? q0 + li : L - 1] : 0;

    auto stage = [&](int buf, int t) {                          // tile t: K rows [32][NP * 128] and Vt columns [VR][32] -> LDS
        char *sk = smem + buf * G::BUF, *sv = sk + G::K_BYTES;
        const int k0 = t * AKT;
        constexpr int KCH = NP * 16;                            // 16-byte chunks per K row
#pragma unroll
        for (int i = 0; i < AKT * KCH / 256; ++i) {
            const int c = tid + i * 256, r = c / KCH, ch = c % KCH;
            const int key = k0 + r < L ? k0 + r : L - 1;
            *(f16x8 *)(sk + r * G::KROW + ch * 16) = *(const f16x8 *)(Kb + (int64_t)key * ldq + ch * 8);
        }
        // Vt: VR rows x 4 chunks (ldv is a multiple of 32: whole chunks are addressable)
        for (int c = tid; c < G::VR * 4; c += 256) {
            const int r = c >> 2, ch = c & 3;
            *(f16x8 *)(sv + r * G::VROW + ch * 16) = *(const f16x8 *)(Vb + (int64_t)r * ldv + k0 + ch * 8);
        }
    };

    f32x16 o[NVB];
#pragma unroll
    for (int bb = 0; bb < NVB; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[bb][r] = 0.f;
    float m = -1e30f, l = 0.f;                                  // running max (log2 units) and sum of this lane's query column
    const float sc = 1.4426950408889634f * 0.08838834764831845f; // log2 e / sqrt(128)
    const int nt = (L + AKT - 1) / AKT;
    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();                                        // tile t is in buffer t & 1; buffer (t + 1) & 1 is free
        if (t + 1 < nt) stage((t + 1) & 1, t + 1);
        const char *sk = smem + (t & 1) * G::BUF, *sv = sk + G::K_BYTES;
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f16x8 kh = *(const f16x8 *)(sk + kperm * G::KROW + (ks * 16 + lh * 8) * 2);
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qf[0][ks], st, 0, 0, 0);
            if constexpr (SPLIT) {
                const f16x8 kl = *(const f16x8 *)(sk + kperm * G::KROW + (AD + ks * 16 + lh * 8) * 2);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qf[1][ks], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qf[0][ks], st, 0, 0, 0);
            }
        }
        // register r of this lane: key k0 + 16 (r >> 3) + 8 lh + (r & 7) (tools/mfma_layout.py); mask, tile max
        float tmax = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * AKT + 16 * (r >> 3) + 8 * lh + (r & 7);
            float s = st[r] * sc;
            if (rg && key < L && rg[key] != my_region) s += -100.f * 1.4426950408889634f;      // the reference adds -100 to the logit
            if (key >= L) s = -1e30f;
            st[r] = s;
            tmax = fmaxf(tmax, s);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mn = fmaxf(m, tmax), corr = __builtin_amdgcn_exp2f(m - mn);
        float psum = 0.f;
        f16x8 pf[NPV][2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(st[r] - mn);
            const f16 ph = (f16)p;
            pf[0][r >> 3][r & 7] = ph;
            if constexpr (SPV) {
                const f16 pl = (f16)(p - (float)ph);
                pf[1][r >> 3][r & 7] = pl;
                psum += (float)ph + (float)pl;                  // the sum of what P V uses
            } else {
                psum += (float)ph;
            }
        }
        psum += __shfl_xor(psum, 32);
        l = l * corr + psum;
        m = mn;
#pragma unroll
        for (int bb = 0; bb < NVB; ++bb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[bb][r] *= corr;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x8 vh = *(const f16x8 *)(sv + (bb * 32 + li) * G::VROW + (s * 16 + lh * 8) * 2);
                o[bb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pf[0][s], o[bb], 0, 0, 0);
                if constexpr (SPV) {
                    const f16x8 vl = *(const f16x8 *)(sv + ((NVB + bb) * 32 + li) * G::VROW + (s * 16 + lh * 8) * 2);
                    o[bb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pf[1][s], o[bb], 0, 0, 0);
                    o[bb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, pf[0][s], o[bb], 0, 0, 0);
                }
            }
        }
    }
    if (q0 + li < L) {
        const float inv = 1.f / l;
        float *dst = O + ((int64_t)b * L + q0 + li) * (NVB * 32);
#pragma unroll
        for (int bb = 0; bb < NVB; ++bb)
#pragma unroll
            for (int r = 0; r < 16; r += 4) {                    // registers r .. r + 3 are 4 consecutive d of this query
                f32x4 v4 = {o[bb][r] * inv, o[bb][r + 1] * inv, o[bb][r + 2] * inv, o[bb][r + 3] * inv};
                *(f32x4 *)(dst + bb * 32 + 8 * (r >> 2) + 4 * lh) = v4;
            }
    }
}

template <bool SPLIT, bool SPV, int NVB>
int launch_t(hipStream_t s, const f16 *Q, const f16 *K, const f16 *Vt, const int8_t *region, float *O, int B, int L, int ldq, int ldv,
             int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int kxor, int nreg) {
    using G = AttnGeom<SPLIT, SPV, NVB>;
    auto kern = attn128_kernel<SPLIT, SPV, NVB>;
    static bool attr_set = false;
    if (!attr_set) {
        PB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G::BUF));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((L + AQ - 1) / AQ, B), dim3(256), 2 * G::BUF, s, Q, K, Vt, region, O, L, ldq, ldv, q_bstride, k_bstride, v_bstride, kxor, nreg);
    PB_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// Q, K [B, L, 128] fp16; Vt [B, 128, ldv] fp16 (ldv >= round_up(L, 32), a multiple of 32; columns >= L may hold anything finite);
// region [B, L] int8 or null (a key whose region id differs from the query's gets -100 on its logit: the shifted-window mask);
// O [B, L, 128] fp32.
int launch_attention128(hipStream_t s, const f16 *Q, const f16 *K, const f16 *Vt, const int8_t *region, float *O, int B, int L, int ldv) {
    PB_CHECK(Q && K && Vt && O && B > 0 && L > 0 && ldv % 32 == 0 && ldv >= (L + 31) / 32 * 32, PB_ERR_ARG, "attention128: bad arguments");
    return launch_t<false, false, 4>(s, Q, K, Vt, region, O, B, L, AD, ldv, (int64_t)L * AD, (int64_t)L * AD, (int64_t)AD * ldv, 0, B);
}

// The general form (kernels.h Attn128Args).
int launch_attention128x(hipStream_t s, const Attn128Args &a) {
    PB_CHECK(a.Q && a.K && a.Vt && a.O && a.B > 0 && a.L > 0 && a.ldv % 32 == 0 && a.ldv >= (a.L + 31) / 32 * 32 && (a.vcols == 128 || a.vcols == 32) &&
                 (!a.region || a.nreg > 0) && (a.kxor == 0 || a.B % (2 * a.kxor) == 0),
             PB_ERR_ARG, "attention128x: bad arguments");
    const int ldq = a.ldq ? a.ldq : (a.split ? 2 * AD : AD), nr = a.region ? a.nreg : 1;
    const int64_t qb = a.q_bstride ? a.q_bstride : (int64_t)a.L * ldq, kb = a.k_bstride ? a.k_bstride : (int64_t)a.L * ldq;
    const int spv = a.split && !a.pv_single;
    const int64_t vb = a.v_shared ? 0 : (a.v_bstride ? a.v_bstride : (int64_t)(spv ? 2 : 1) * a.vcols * a.ldv);
#define PB_ATT(S, P, N) launch_t<S, P, N>(s, a.Q, a.K, a.Vt, a.region, a.O, a.B, a.L, ldq, a.ldv, qb, kb, vb, a.kxor, nr)
    if (a.split && spv) return a.vcols == 128 ? PB_ATT(true, true, 4) : PB_ATT(true, true, 1);
    if (a.split) return a.vcols == 128 ? PB_ATT(true, false, 4) : PB_ATT(true, false, 1);
    return a.vcols == 128 ? PB_ATT(false, false, 4) : PB_ATT(false, false, 1);
#undef PB_ATT
}
