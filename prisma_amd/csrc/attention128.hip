// Single-head attention over 128-wide heads: O[b][q] = softmax_k(Q[b][q] . K[b][k] / sqrt(128) + mask(q, k)) V[b][k] - the
// attention of the reference's GMFlow transformer (bands/gmflow/transformer.py:8-15 full, :47-101 split windows with the shifted-window
// mask of :18-44) and, with V = coordinates / flow, of its global matching and flow propagation (matching.py:7-42, transformer.py:316-337).
// First building block of the flow_gmflow band; so far it is only reachable through the pb_op_attention128 test hook.
//
// Flash style, in the layouts tools/mfma_layout.py models and tests/test_mfma_layout_cpu.py runs in numpy:
//   * a wave owns 32 queries (Q fragments of the 8 k-steps in registers), a workgroup 4 waves = 128 queries of one batch element;
//   * per tile of 32 keys, S^T[key][q] = K Q^T with the K rows in swap_bits23 order: a lane holds 16 scores of ONE query column (row
//     max / sum in-lane plus one exchange with lane ^ 32) and registers 8 s .. 8 s + 7 are the B fragment of k-step s of
//     O^T[d][q] += Vt[d][key] P^T[key][q] (four 32-row blocks of d);
//   * K and Vt tiles go through the LDS (plain loads + ds_write; rows padded to 272 / 80 bytes: conflict-free 16-byte fragment reads),
//     double buffered, one barrier per tile;
//   * scores and the running max live in log2 units (Q . K scaled by log2 e / sqrt(128)); statistics, O and the output are fp32.
// Not tuned: no LDS-DMA, no matrix-pipe softmax tricks (attention.hip has those for the 64-wide ViT heads).
#include "common.h"
#include "kernels.h"
#include "../../include/prisma_bands.h"

namespace {

constexpr int AQ = 128, AKT = 32, AD = 128;                     // queries per workgroup, keys per tile, head dim
constexpr int KROW = AD * 2 + 16, VROW = AKT * 2 + 16;          // LDS row strides in bytes (272, 80)
constexpr int K_BYTES = AKT * KROW, V_BYTES = AD * VROW, BUF = K_BYTES + V_BYTES;

__global__ __launch_bounds__(256) void attn128_kernel(const f16 *__restrict__ Q, const f16 *__restrict__ K, const f16 *__restrict__ Vt,
                                                      const int8_t *__restrict__ region, float *__restrict__ O, int L, int ldv) {
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * AQ + wave * 32;
    const f16 *Qb = Q + (int64_t)b * L * AD, *Kb = K + (int64_t)b * L * AD, *Vb = Vt + (int64_t)b * AD * ldv;
    const int8_t *rg = region ? region + (int64_t)b * L : nullptr;
    const int kperm = (li & 19) | ((li & 4) << 1) | ((li & 8) >> 1);

    // Q fragments (B operand of S^T): query q0 + li, k-steps 0..7
    f16x8 qf[8];
    {
        const int q = q0 + li < L ? q0 + li : L - 1;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const f16x8 *)(Qb + (int64_t)q * AD + ks * 16 + lh * 8);
    }
    const int my_region = rg ? rg[q0 + li < L ? q0 + li : L - 1] : 0;

    auto stage = [&](int buf, int t) {                          // tile t: K rows [32][128] and Vt columns [128][32] -> LDS
        char *sk = smem + buf * BUF, *sv = sk + K_BYTES;
        const int k0 = t * AKT;
        {   // K: 32 rows x 16 chunks of 16 B = 512 chunks, 2 per thread
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + i * 256, r = c >> 4, ch = c & 15;
                const int key = k0 + r < L ? k0 + r : L - 1;
                *(f16x8 *)(sk + r * KROW + ch * 16) = *(const f16x8 *)(Kb + (int64_t)key * AD + ch * 8);
            }
        }
        {   // Vt: 128 rows x 4 chunks = 512 chunks, 2 per thread (ldv is a multiple of 32: whole chunks are addressable)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tid + i * 256, r = c >> 2, ch = c & 3;
                *(f16x8 *)(sv + r * VROW + ch * 16) = *(const f16x8 *)(Vb + (int64_t)r * ldv + k0 + ch * 8);
            }
        }
    };

    f32x16 o[4];
#pragma unroll
    for (int bb = 0; bb < 4; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[bb][r] = 0.f;
    float m = -1e30f, l = 0.f;                                  // running max (log2 units) and sum of this lane's query column
    const float sc = 1.4426950408889634f * 0.08838834764831845f; // log2 e / sqrt(128)
    const int nt = (L + AKT - 1) / AKT;
    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();                                        // tile t is in buffer t & 1; buffer (t + 1) & 1 is free
        if (t + 1 < nt) stage((t + 1) & 1, t + 1);
        const char *sk = smem + (t & 1) * BUF, *sv = sk + K_BYTES;
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f16x8 kf = *(const f16x8 *)(sk + kperm * KROW + (ks * 16 + lh * 8) * 2);
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], st, 0, 0, 0);
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
        f16x8 pf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(st[r] - mn);
            const f16 ph = (f16)p;
            psum += (float)ph;                                  // the sum of the ROUNDED probabilities, i.e. of what P V uses
            pf[r >> 3][r & 7] = ph;
        }
        psum += __shfl_xor(psum, 32);
        l = l * corr + psum;
        m = mn;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[bb][r] *= corr;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x8 vf = *(const f16x8 *)(sv + (bb * 32 + li) * VROW + (s * 16 + lh * 8) * 2);
                o[bb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[s], o[bb], 0, 0, 0);
            }
        }
    }
    if (q0 + li < L) {
        const float inv = 1.f / l;
        float *dst = O + ((int64_t)b * L + q0 + li) * AD;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[bb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = o[bb][r] * inv;
    }
}

}  // namespace

// Q, K [B, L, 128] fp16; Vt [B, 128, ldv] fp16 (ldv >= round_up(L, 32), a multiple of 32; columns >= L may hold anything finite);
// region [B, L] int8 or null (a key whose region id differs from the query's gets -100 on its logit: the shifted-window mask);
// O [B, L, 128] fp32.
int launch_attention128(hipStream_t s, const f16 *Q, const f16 *K, const f16 *Vt, const int8_t *region, float *O, int B, int L, int ldv) {
    PB_CHECK(Q && K && Vt && O && B > 0 && L > 0 && ldv % 32 == 0 && ldv >= (L + 31) / 32 * 32, PB_ERR_ARG, "attention128: bad arguments");
    hipLaunchKernelGGL(attn128_kernel, dim3((L + AQ - 1) / AQ, B), dim3(256), 0, s, Q, K, Vt, region, O, L, ldv);
    PB_HIP(hipGetLastError());
    return 0;
}
