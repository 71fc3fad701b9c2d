// Single-head attention over 128-wide heads: O[b][q] = softmax_k(Q[b][q] . K[b][k] / sqrt(128) + mask(q, k)) V[b][k] - the
// attention of the reference's GMFlow transformer (bands/gmflow/transformer.py:8-15 full, :47-101 split windows with the shifted-window
// mask of :18-44) and, with V = coordinates / flow, of its global matching and flow propagation (matching.py:7-42, transformer.py:316-337).
//
// Flash style, in the layouts tools/mfma_layout.py models and tests/test_mfma_layout_cpu.py runs in numpy:
//   * a wave owns 32 queries (Q fragments of the 8 k-steps in registers), a workgroup 4 waves = 128 queries of one batch element;
//   * per tile of 32 keys, S^T[key][q] = K Q^T with the K rows in swap_bits23 order: a lane holds 16 scores of ONE query column (row
//     max / sum in-lane plus one exchange with lane ^ 32) and registers 8 s .. 8 s + 7 are the B fragment of k-step s of
//     O^T[d][q] += Vt[d][key] P^T[key][q] (NVB 32-row blocks of d: 4 for a 128-wide V, 1 for the 2-column coordinate / flow V);
//   * K and Vt tiles reach the LDS by LDS-DMA (`global_load_lds`, no staging registers, 16-byte chunks XOR-swizzled at the source so
//     that the unpadded rows give conflict-free 16-byte fragment reads), three buffers, two tiles ahead: a tile's DMAs have a whole
//     tile of MFMAs + softmax to land, and the one wait per tile is a counted `s_waitcnt vmcnt` that leaves the newest stage in flight;
//     the window mask's region ids are copied to the LDS once (a global load inside the loop would wait behind the DMAs);
//   * scores and the running max live in log2 units (Q . K scaled by log2 e / sqrt(128)); statistics, O and the output are fp32.
// SPLIT (the flow_gmflow band's precision mode, DESIGN.md section 7: this network's two softmax stages amplify operand rounding): Q, K and
// V arrive as hi + lo fp16 pairs and P is split in registers, S = q_hi k_hi + q_lo k_hi + q_hi k_lo and O += v_hi p_hi + v_hi p_lo +
// v_lo p_hi in the same fp32 accumulators - three MFMA passes each, operands good to ~22 bits.
// (First version, round 3: plain loads + ds_write through registers, two buffers - every tile exposed a global-load latency before its
// ds_writes; the matrix pipe was busy ~27 % of the time.)
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
    static constexpr int KROW = (SPLIT ? 2 : 1) * AD * 2;        // LDS row strides in bytes (256 / 512, 64): unpadded, chunks swizzled
    static constexpr int VROW = AKT * 2;
    static constexpr int VR = (SPV ? 2 : 1) * NVB * 32;         // Vt rows per tile: [hi rows | lo rows]
    static constexpr int VRS = VR < 64 ? 64 : VR;               // rows staged: at least one 1 KB DMA per wave (the extra rows repeat valid ones)
    static constexpr int K_BYTES = AKT * KROW, V_BYTES = VRS * VROW, BUF = K_BYTES + V_BYTES;
    static constexpr int NK = K_BYTES / 4096, NV = V_BYTES / 4096;      // DMAs per thread and stage
    static constexpr int NBUF = 3 * BUF <= 80 * 1024 ? 3 : 2;   // three buffers where two workgroups still share a CU
    static constexpr int REG_MAX = 8192;                        // region ids kept in the LDS up to this many tokens
};

// Q, K: rows of ldq halfs, the hi part in columns [0, 128), with SPLIT the lo part in [128, 256); batch element b reads queries at
//       Q + b q_bstride and keys / values at index b ^ kxor (K + (b ^ kxor) k_bstride: cross attention between the two frames of a pair).
// Vt:   [.., VR, ldv] fp16 (v_bstride = 0 shares one V between all batch elements): rows [0, NVB * 32) hi, then lo.
// region: [nreg, L] int8 or null; batch element b uses row b % nreg.
// O:    [B, L, NVB * 32] fp32.
// RMODE: 0 no region mask, 1 region ids in the LDS (L <= REG_MAX), 2 read from global memory inside the loop (any L; the compiler's waits for
// those loads also wait for the DMAs in flight - slower)
#define A128_BAR()                              \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        asm volatile("s_barrier" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

template <bool SPLIT, bool SPV, int NVB, int RMODE>
__global__ __launch_bounds__(256, 2) void attn128_kernel(const f16 *__restrict__ Q, const f16 *__restrict__ K, const f16 *__restrict__ Vt,
                                                      const int8_t *__restrict__ region, float *__restrict__ O, int L, int ldq, int ldv,
                                                      int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int kxor, int nreg) {
    using G = AttnGeom<SPLIT, SPV, NVB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * AQ + wave * 32;
    const f16 *Qb = Q + (int64_t)b * q_bstride, *Kb = K + (int64_t)(b ^ kxor) * k_bstride, *Vb = Vt + (int64_t)(b ^ kxor) * v_bstride;
    const int8_t *rg = RMODE ? region + (int64_t)(b % nreg) * L : nullptr;
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
    int my_region = rg ? rg[q0 + li < L ? q0 + li : L - 1] : 0;
    // Pin the loads above here: the compiler's waitcnt pass would otherwise wait for them at their first use INSIDE the loop, with a count
    // that (loads return in order) also drains the DMA stage the loop is trying to keep in flight.
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[p][ks]));
    asm volatile("" : "+v"(my_region));

    // tile t -> buffer `buf`: K rows [32][NP * 128 halfs] and Vt rows [VRS][32 keys], 1 KB (64 lanes x 16 B) per DMA.  A lane's LDS slot is
    // (row, physical chunk p); it fetches logical chunk p ^ f(row) - the same XOR the fragment reads apply (K: low 4 bits of the row on the
    // low 4 bits of the chunk, Vt: (row >> 2) & 3), which spreads the 16 rows a 16-lane group reads at one logical chunk over all 16
    // 16-byte columns of the LDS.
    // (buffer path: the per-lane byte offsets are loop invariant, the tile advance is a wave-uniform scalar offset, and key rows past L
    // fall outside the resource and read zeros - their scores are masked below)
    const __amdgpu_buffer_rsrc_t rsK = make_rsrc(Kb, (unsigned)((int64_t)L * ldq * 2)), rsV = make_rsrc(Vb, (unsigned)((int64_t)G::VR * ldv * 2));
    int k_voff[G::NK], v_voff[G::NV > 0 ? G::NV : 1];
#pragma unroll
    for (int i = 0; i < G::NK; ++i) {
        const int o = (i * 4 + wave) * 1024 + lane * 16;
        const int r = o / G::KROW, pch = (o % G::KROW) >> 4;
        k_voff[i] = (r * ldq + ((pch & ~15) | ((pch ^ r) & 15)) * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < G::NV; ++i) {
        const int o = (i * 4 + wave) * 1024 + lane * 16;
        const int r = o / G::VROW, pch = (o % G::VROW) >> 4;
        v_voff[i] = ((G::VRS == G::VR ? r : r % G::VR) * ldv + (pch ^ ((r >> 2) & 3)) * 8) * 2;
    }
    auto stage = [&](int buf, int t) {
        char *sk = smem + buf * G::BUF, *sv = sk + G::K_BYTES;
        const int k0 = t * AKT;
#pragma unroll
        for (int i = 0; i < G::NK; ++i) glds16_buf(rsK, k_voff[i], k0 * ldq * 2, sk + (i * 4 + wave) * 1024);
#pragma unroll
        for (int i = 0; i < G::NV; ++i) glds16_buf(rsV, v_voff[i], k0 * 2, sv + (i * 4 + wave) * 1024);
    };
    // region ids of all keys in the LDS (behind the tile buffers)
    int8_t *sreg = (int8_t *)(smem + G::NBUF * G::BUF);
    if constexpr (RMODE == 1) {
        for (int i = tid; i < (L + 31) / 32 * 32; i += 256) sreg[i] = rg[i < L ? i : L - 1];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the loop's barriers are bare s_barrier instructions)
    }

    f32x16 o[NVB];
#pragma unroll
    for (int bb = 0; bb < NVB; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[bb][r] = 0.f;
    float m = -1e30f, l = 0.f;                                  // running max (log2 units) and sum of this lane's query column
    const float sc = 1.4426950408889634f * 0.08838834764831845f; // log2 e / sqrt(128)
    const int nt = (L + AKT - 1) / AKT;
    constexpr int AHEAD = G::NBUF - 1;                          // tiles in flight beyond the current one
    stage(0, 0);
    if (AHEAD > 1 && nt > 1) stage(1, 1);
    int bcur = 0;                                               // buffer of tile t
    for (int t = 0; t < nt; ++t) {
        // tile t has landed when at most the newer stage's DMAs are outstanding (loads return in order; nothing else is in flight)
        if (AHEAD > 1 && t + 1 < nt) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NK + G::NV) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        A128_BAR();                                             // ... for every wave's share; the buffer of tile t - 1 is free.  A bare barrier:
                                                                // __syncthreads() would drain vmcnt and with it the stage in flight
        if (t + AHEAD < nt) {
            int bn = bcur + AHEAD;
            bn = bn >= G::NBUF ? bn - G::NBUF : bn;
            stage(bn, t + AHEAD);
        }
        const char *sk = smem + bcur * G::BUF, *sv = sk + G::K_BYTES;
        bcur = bcur + 1 == G::NBUF ? 0 : bcur + 1;
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f16x8 kh = *(const f16x8 *)(sk + kperm * G::KROW + (((2 * ks + lh) ^ kperm) & 15) * 16);
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qf[0][ks], st, 0, 0, 0);
            if constexpr (SPLIT) {
                const f16x8 kl = *(const f16x8 *)(sk + kperm * G::KROW + 256 + (((2 * ks + lh) ^ kperm) & 15) * 16);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qf[1][ks], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qf[0][ks], st, 0, 0, 0);
            }
        }
        // register r of this lane: key k0 + 16 (r >> 3) + 8 lh + (r & 7) (tools/mfma_layout.py); mask, tile max
        float tmax = -1e30f;
        unsigned long long rw[2] = {0ull, 0ull};                // region ids of this lane's two runs of 8 keys
        if constexpr (RMODE == 1) {
            rw[0] = *(const unsigned long long *)(sreg + t * AKT + 8 * lh);
            rw[1] = *(const unsigned long long *)(sreg + t * AKT + 16 + 8 * lh);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * AKT + 16 * (r >> 3) + 8 * lh + (r & 7);
            float s = st[r] * sc;
            // the reference adds -100 to the logit of a key in another region
            if constexpr (RMODE == 1) s += (int)(signed char)(rw[r >> 3] >> (8 * (r & 7))) != my_region ? -100.f * 1.4426950408889634f : 0.f;
            if constexpr (RMODE == 2) { if (key < L && rg[key] != my_region) s += -100.f * 1.4426950408889634f; }
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
                const int vsw = (li >> 2) & 3;                  // rows bb * 32 + li and (NVB + bb) * 32 + li share (row >> 2) & 3
                const f16x8 vh = *(const f16x8 *)(sv + (bb * 32 + li) * G::VROW + (((2 * s + lh) ^ vsw) * 16));
                o[bb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pf[0][s], o[bb], 0, 0, 0);
                if constexpr (SPV) {
                    const f16x8 vl = *(const f16x8 *)(sv + ((NVB + bb) * 32 + li) * G::VROW + (((2 * s + lh) ^ vsw) * 16));
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

template <bool SPLIT, bool SPV, int NVB, int RMODE>
int launch_r(hipStream_t s, const f16 *Q, const f16 *K, const f16 *Vt, const int8_t *region, float *O, int B, int L, int ldq, int ldv,
             int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int kxor, int nreg) {
    using G = AttnGeom<SPLIT, SPV, NVB>;
    auto kern = attn128_kernel<SPLIT, SPV, NVB, RMODE>;
    static bool attr_set = false;
    if (!attr_set) {
        PB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::NBUF * G::BUF + G::REG_MAX));
        attr_set = true;
    }
    const int smem_bytes = G::NBUF * G::BUF + (RMODE == 1 ? (L + 31) / 32 * 32 : 0);
    hipLaunchKernelGGL(kern, dim3((L + AQ - 1) / AQ, B), dim3(256), smem_bytes, s, Q, K, Vt, region, O, L, ldq, ldv, q_bstride, k_bstride, v_bstride, kxor, nreg);
    PB_HIP(hipGetLastError());
    return 0;
}

template <bool SPLIT, bool SPV, int NVB>
int launch_t(hipStream_t s, const f16 *Q, const f16 *K, const f16 *Vt, const int8_t *region, float *O, int B, int L, int ldq, int ldv,
             int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int kxor, int nreg) {
    using G = AttnGeom<SPLIT, SPV, NVB>;
#define PB_A128(R) launch_r<SPLIT, SPV, NVB, R>(s, Q, K, Vt, region, O, B, L, ldq, ldv, q_bstride, k_bstride, v_bstride, kxor, nreg)
    if (!region) return PB_A128(0);
    if (L <= G::REG_MAX) return PB_A128(1);
    return PB_A128(2);
#undef PB_A128
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
