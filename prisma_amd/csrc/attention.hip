// Fused softmax(Q K^T) V for the DINOv2 blocks (reference dinov2/layers/attention.py:49-62:
// q * 64^-0.5, q @ k^T, softmax, @ v) on gfx950 - flash style, the N x N score matrix never
// leaves registers.  head_dim is 64 for every encoder size.
//
// Layouts (written by the qkv GEMM epilogue, gemm.hip EPI_QKV):
//   Q, K : [b, head, ntp, 64] fp16, Q already multiplied by PB_QSCALE = 64^-0.5 * log2(e): scores are in log2 units
//   Vt   : [b, head, 64, ntp] fp16 (V transposed: keys contiguous)
//   O    : [b * ntp, heads * 64] fp16 row-major (the proj GEMM's A operand)
//
// Workgroup = NW waves (8 by default) = 32 * NW query rows of one (b, head); each wave owns NQB blocks of 32 query rows.
// Per 64-key tile:
//   S^T[key][q] = K Q^T  with v_mfma_f32_32x32x16_f16 (A = K rows, B = Q rows): a lane then holds
//                 32 scores of ONE query column, so row max / row sum are in-lane plus one
//                 exchange with lane ^ 32.
//   The K rows of a 32-key sub-tile are fed in the order swap_bits23(row): with that order
//   accumulator registers 8j..8j+7 of a lane are 8 CONSECUTIVE keys, i.e. exactly the B-operand
//   fragment of the second matmul - P never moves between lanes.
//   O^T[d][q] += Vt[d][key] P^T[key][q]  (A = Vt rows, one ds_read_b128 per fragment).
//   Softmax: the reference point m and the padding mask ride on a fifth k-step of the S MFMA chain, the
//   steady state per score is one v_exp_f32, half a v_cvt_pk_f16_f32 and half a v_dot2_f32_f16 (see the loop).
// K and Vt tiles go L2 -> LDS with buffer_load_dwordx4 ... lds, double buffered, XOR-swizzled
// through the source address exactly like gemm.hip.
// Measured (tools/attn_bench.py, B = 32, 16 heads, 2443 tokens, rocprofv3 SQ counters in profiles/): 0.875 ms = 890 TF.
// Timing-only ablations of the same loop: MFMAs + ds_reads alone 0.60 ms, + LDS-DMA 0.71 ms, + softmax VALU 0.875 ms;
// VALU alone 0.52 ms; barriers, LDS read latency (PF), occupancy 2 / 3 / 4 waves per SIMD and NQB = 2 all within 2 %.
#include "common.h"

#include <stdlib.h>

namespace {

struct AttnArgs {
    const f16 *q, *k, *vt;
    f16 *o;
    int ntp, ntok, heads, ldo, nq, nb;
    int o8_off;
    float o8_scale;
};

// NQB query blocks of 32 rows per wave: with NQB = 2 a K / Vt fragment feeds two MFMAs on independent accumulators and
// the LDS traffic per flop halves; measured no faster than NQB = 1 (which runs 4 waves per SIMD instead of 2).
// ABL != 0 are timing-only ablations (wrong results): 1 no v_exp, 2 no MFMAs, 3 no softmax VALU at all;
// 4 = 3 + no LDS-DMA, 5 = 4 + no barrier, 6 = 3 + no ds_reads (fragments from registers)
// PF: all K fragments of a tile are read from the LDS before its first MFMA and all Vt fragments before the softmax, so
// the matrix pipe never waits for an LDS round trip (without it hipcc keeps one fragment in flight to save registers).
// NW waves per workgroup (4 or 8): 32 * NQB * NW query rows share one K / Vt stream, so the LDS-DMA instructions per
// wave and tile (the issue cost of which is ~15 % of the loop at NW = 4) go down as 8 / NW.
// NBUF = 3 (round 5, the schedule of attention128.hip): three K / Vt buffers, tile t + 2 requested at the top of tile t, ONE counted
// `s_waitcnt vmcnt` per tile that leaves the newest stage in flight, and a bare `s_barrier` (__syncthreads() drains vmcnt and with it
// the stage the loop is trying to keep in flight).  NBUF = 2 is the round-1 loop: vmcnt(0) + __syncthreads() per tile, one tile ahead.
#define ATT_BAR()                               \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        asm volatile("s_barrier" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

template <int NQB, int OCC, int ABL = 0, bool PF = false, int NW = 4, int NBUF = 2>
__global__ __launch_bounds__(64 * NW, OCC) void attnq_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[NBUF * 16384];   // NBUF x (K tile 8 KB + Vt tile 8 KB)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int nq = p.nq, nbh = p.heads * p.nb;                       // nq = q-blocks of 128 * NQB rows
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bhi = (slot / nq) * 8 + xcd, qblk = slot % nq;
    if (bhi >= nbh) return;
    const int b = bhi / p.heads, head = bhi - b * p.heads;
    const int64_t bh = bhi;
    const f16 *Q = p.q + bh * p.ntp * 64;
    const f16 *K = p.k + bh * p.ntp * 64;
    const f16 *Vt = p.vt + bh * 64 * p.ntp;

    int qrow[NQB];
    f16x8 qf[NQB][4];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        qrow[qb] = qblk * (32 * NW * NQB) + wave * (32 * NQB) + qb * 32 + li;
        const int qrc = qrow[qb] < p.ntp ? qrow[qb] : p.ntp - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[qb][s] = *(const f16x8 *)(Q + (int64_t)qrc * 64 + 16 * s + 8 * lh);
    }
    if constexpr (NBUF == 3) {
        // pin the Q loads here: the compiler's wait for them at their first use INSIDE the loop would (loads return in order) also
        // drain the DMA stages the loop keeps in flight
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[qb][s]));
    }

    constexpr int CH = 8 / NW;                               // 16-byte chunks of K (and of Vt) per thread and tile
    const int srow = tid >> 3;                               // 0 .. 8 NW - 1 (+ 8 NW per further chunk)
    const int cg = (tid & 7) ^ ((tid >> 4) & 7);
    const __amdgpu_buffer_rsrc_t rsK = make_rsrc(K, (unsigned)(p.ntp * 64 * 2)), rsV = make_rsrc(Vt, (unsigned)(64 * p.ntp * 2));
    int kvo[CH], vvo[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int r = srow + 8 * NW * i;
        kvo[i] = r * 128 + cg * 16;
        vvo[i] = (r * p.ntp + cg * 8) * 2;
    }
    auto stage = [&](int buf, int t) {
        char *sK = smem + buf * 16384 + wave * 1024;
        char *sV = sK + 8192;
        const int key0 = t * 64;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            glds16_buf(rsK, kvo[i], key0 * 128, sK + i * (NW * 1024));
            glds16_buf(rsV, vvo[i], key0 * 2, sV + i * (NW * 1024));
        }
    };

    const int fsw = (li >> 1) & 7;
    const int kperm = (li & 19) | ((li & 4) << 1) | ((li & 8) >> 1);
    const int fswk = (kperm >> 1) & 7;

    f32x16 oacc[NQB][2];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][d][r] = 0.f;
    float mrun[NQB], lrun[NQB];
    f16x8 qm[NQB], kone;
#pragma unroll
    for (int j = 0; j < 8; ++j) kone[j] = (f16)1.f;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        mrun[qb] = 0.f; lrun[qb] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) qm[qb][j] = (f16)0.f;
        qm[qb][1] = lh == 0 ? (f16)-30000.f : (f16)0.f;
    }
    const f16x2 one2 = {(f16)1.f, (f16)1.f};

    const int nt = (p.ntok + 63) >> 6;
    stage(0, 0);
    if constexpr (NBUF == 3) stage(1, nt > 1 ? 1 : 0);
    int bcur = 0;                                            // NBUF = 3: buffer of tile t
    for (int t = 0; t < nt; ++t) {
        const char *sK;
        if constexpr (NBUF == 3) {
            // tile t has landed when at most the newer stage's 2 CH DMAs are outstanding (nothing else is in flight); the stage issued
            // below overwrites the buffer tile t - 1 was read from, which every wave has left once it is past the barrier
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CH) : "memory");
            ATT_BAR();
            int bn = bcur + 2;
            bn = bn >= 3 ? bn - 3 : bn;
            stage(bn, t + 2 < nt ? t + 2 : nt - 1);          // (past the end: a valid tile again, so the wait count stays uniform)
            sK = smem + bcur * 16384;
            bcur = bcur == 2 ? 0 : bcur + 1;
        } else {
            if constexpr (ABL != 5) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            if constexpr (ABL != 4 && ABL != 5) stage((t + 1) & 1, t + 1 < nt ? t + 1 : t);
            sK = smem + (t & 1) * 16384;
        }
        const char *sV = sK + 8192;

        // ---- S^T = K Q^T - m, keys >= ntok pushed to -30000 (2^-30000 = 0) by the same extra k-step: the K-side
        //      fragment holds 1 in k-slot 0 (times -m in the Q-side fragment) and [key >= ntok] in k-slot 1 (times
        //      -30000), so neither the reference point nor the padding mask costs a VALU instruction or a branch ----
        f16x8 kfp[2][4];
        if constexpr (PF) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kfp[kt][ks] = *(const f16x8 *)(sK + (kt * 32 + kperm) * 128 + (((2 * ks + lh) ^ fswk) * 16));
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x16 s[NQB][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f16x8 kx = kone;
            kx[1] = (t * 64 + kt * 32 + kperm >= p.ntok) ? (f16)1.f : (f16)0.f;
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[qb][kt][r] = 0.f;
                s[qb][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kx, qm[qb], s[qb][kt], 0, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f16x8 kf;
                if constexpr (PF) kf = kfp[kt][ks];
                else if constexpr (ABL == 6) kf = qf[0][ks ^ 1];
                else kf = *(const f16x8 *)(sK + (kt * 32 + kperm) * 128 + (((2 * ks + lh) ^ fswk) * 16));
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    if constexpr (ABL == 2) s[qb][kt][ks] += (float)kf[0] * (float)qf[qb][ks][0];
                    else s[qb][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][ks], s[qb][kt], 0, 0, 0);
                }
            }
        }
        f16x8 vfp[4][2];
        if constexpr (PF) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int d = 0; d < 2; ++d) vfp[j][d] = *(const f16x8 *)(sV + (d * 32 + li) * 128 + (((2 * j + lh) ^ fsw) * 16));
            __builtin_amdgcn_sched_barrier(0);
        }
        float ps[NQB];
        f16x8 pf[NQB][4];
        auto probs = [&](int qb) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g = 0; g < 16; g += 2) {
                    f16x2 h;
                    if constexpr (ABL == 1 || ABL >= 3) h = f16x2{(f16)s[qb][kt][g], (f16)s[qb][kt][g + 1]};
                    else h = f16x2{(f16)__builtin_amdgcn_exp2f(s[qb][kt][g]), (f16)__builtin_amdgcn_exp2f(s[qb][kt][g + 1])};
                    if constexpr (ABL < 3) {
                        if (kt == 0) a0 = __builtin_amdgcn_fdot2(h, one2, a0, false);
                        else a1 = __builtin_amdgcn_fdot2(h, one2, a1, false);
                    }
                    pf[qb][kt * 2 + (g >> 3)][g & 7] = h[0];
                    pf[qb][kt * 2 + (g >> 3)][(g & 7) + 1] = h[1];
                }
            ps[qb] = a0 + a1;
        };
        bool bad = t == 0;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            probs(qb);
            bad |= !(ps[qb] <= 1024.f);
        }
        if (__any(bad)) {
            asm volatile("" ::: "memory");               // rarely taken: first tile, or a row max outgrew m by ~10 octaves
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                float mx = s[qb][0][0];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int g = 0; g < 16; ++g) mx = fmaxf(mx, s[qb][kt][g]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float rel = t == 0 ? mx : fmaxf(mx, 0.f);
                const float m16 = (float)(f16)fminf(fmaxf(mrun[qb] + rel, -6.0e4f), 6.0e4f);
                const float d = m16 - mrun[qb];
                const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f(-d);
                mrun[qb] = m16;
                lrun[qb] *= alpha;
#pragma unroll
                for (int dd = 0; dd < 2; ++dd)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[qb][dd][r] *= alpha;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int g = 0; g < 16; ++g) s[qb][kt][g] -= d;
                qm[qb][0] = lh == 0 ? (f16)(-m16) : (f16)0.f;
                probs(qb);
            }
        }
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) lrun[qb] += ps[qb];

        // ---- O^T += Vt P^T ----
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                f16x8 vf;
                if constexpr (PF) vf = vfp[j][d];
                else if constexpr (ABL == 6) vf = qf[0][j];
                else vf = *(const f16x8 *)(sV + (d * 32 + li) * 128 + (((2 * j + lh) ^ fsw) * 16));
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    if constexpr (ABL == 2) oacc[qb][d][j] += (float)vf[0] * (float)pf[qb][j][0];
                    else oacc[qb][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[qb][j], oacc[qb][d], 0, 0, 0);
                }
            }
    }

    if constexpr (NBUF == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the two stages past the end
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const float ltot = lrun[qb] + __shfl_xor(lrun[qb], 32);
        const float inv = 1.f / ltot;
        if (qrow[qb] < p.ntok) {
            f16 *orow = p.o + ((int64_t)b * p.ntp + qrow[qb]) * p.ldo + head * 64;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f16x4 r;
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[j] = (f16)(oacc[qb][d][g * 4 + j] * inv);
                    *(f16x4 *)(orow + d * 32 + 8 * g + 4 * lh) = r;
                    if (p.o8_off)       // fp8 copy of the row after its fp16 part: the A operand of proj's MX correction segment
                        *(int *)((char *)(orow - head * 64) + p.o8_off + head * 64 + d * 32 + 8 * g + 4 * lh) =
                            pb_fp8x4((float)r[0] * p.o8_scale, (float)r[1] * p.o8_scale, (float)r[2] * p.o8_scale, (float)r[3] * p.o8_scale);
                }
        }
    }
}

}  // namespace

int launch_attention(hipStream_t stream, const f16 *q, const f16 *k, const f16 *vt, f16 *o, int B, int heads,
                     int ntp, int ntok, int ldo, int variant, int o8_off, float o8_scale) {
    static int env_variant = -1;
    if (env_variant < 0) { const char *e = getenv("PB_ATTN_VARIANT"); env_variant = e ? atoi(e) : 0; }
    if (variant <= 0) variant = env_variant;
    // a small batch does not fill the chip with 8-wave workgroups (one 720p frame: 16 heads x 10 blocks of 256 rows = 160): the 4-wave
    // geometry (128 rows per workgroup, same per-wave arithmetic - bit-identical output) puts twice as many on it
    if (variant == 0 && (int64_t)((ntok + 255) / 256) * ((B * heads + 7) / 8) * 8 < 384) variant = 2;
    // variants (tools/attn_bench.py): 0 default = 8 waves x 1 block; 2 = 4 waves x 1 block; 3 = 4 waves x 2 blocks;
    // 4 = 4 waves x 1 block with fragment prefetch; 5 = 8 waves x 2 blocks; 11 .. 16 = ablations 1 .. 6 of variant 2;
    // 6 = 8 waves x 1 block on three buffers (round 5), 7 = 8 waves x 2 blocks on three buffers; 21 / 23 = ablations 1 / 3 of variant 6
    const int nqb = variant == 3 || variant == 5 || variant == 7 ? 2 : 1;
    const int nw = variant == 0 || variant == 1 || variant == 5 || variant == 6 || variant == 7 || variant == 21 || variant == 23 ? 8 : 4;
    const int nq = (ntok + 32 * nw * nqb - 1) / (32 * nw * nqb);
    AttnArgs a{q, k, vt, o, ntp, ntok, heads, ldo, nq, B, o8_off, o8_scale};
    dim3 grid(8 * nq * ((B * heads + 7) / 8));
    // template arguments: NQB, minimum workgroups per CU the register allocation must allow, ablation, prefetch, NW
    // the shipped library carries the two geometries the engines use; the experiment / ablation variants of tools/attn_bench.py (3-7, 11-16, 21, 23:
    // wrong results for the ablations) are compiled with -DPB_DIAG only (make EXTRA=-DPB_DIAG BUILD=build_diag LIB=../libprisma_bands_diag.so)
    switch (variant) {
    case 2: hipLaunchKernelGGL((attnq_kernel<1, 4, 0, false, 4>), grid, dim3(256), 0, stream, a); break;
#ifdef PB_DIAG
    case 3: hipLaunchKernelGGL((attnq_kernel<2, 2, 0, false, 4>), grid, dim3(256), 0, stream, a); break;
    case 4: hipLaunchKernelGGL((attnq_kernel<1, 3, 0, true, 4>), grid, dim3(256), 0, stream, a); break;
    case 5: hipLaunchKernelGGL((attnq_kernel<2, 1, 0, false, 8>), grid, dim3(512), 0, stream, a); break;
    // (the second launch-bound argument is waves per SIMD: 4 = two 8-wave workgroups per CU, i.e. at most 128 registers)
    case 6: hipLaunchKernelGGL((attnq_kernel<1, 4, 0, false, 8, 3>), grid, dim3(512), 0, stream, a); break;
    case 7: hipLaunchKernelGGL((attnq_kernel<2, 2, 0, false, 8, 3>), grid, dim3(512), 0, stream, a); break;
    case 21: hipLaunchKernelGGL((attnq_kernel<1, 4, 1, false, 8, 3>), grid, dim3(512), 0, stream, a); break;
    case 23: hipLaunchKernelGGL((attnq_kernel<1, 4, 3, false, 8, 3>), grid, dim3(512), 0, stream, a); break;
    case 11: hipLaunchKernelGGL((attnq_kernel<1, 4, 1, false, 4>), grid, dim3(256), 0, stream, a); break;
    case 12: hipLaunchKernelGGL((attnq_kernel<1, 4, 2, false, 4>), grid, dim3(256), 0, stream, a); break;
    case 13: hipLaunchKernelGGL((attnq_kernel<1, 4, 3, false, 4>), grid, dim3(256), 0, stream, a); break;
    case 14: hipLaunchKernelGGL((attnq_kernel<1, 4, 4, false, 4>), grid, dim3(256), 0, stream, a); break;
    case 15: hipLaunchKernelGGL((attnq_kernel<1, 4, 5, false, 4>), grid, dim3(256), 0, stream, a); break;
    case 16: hipLaunchKernelGGL((attnq_kernel<1, 4, 6, false, 4>), grid, dim3(256), 0, stream, a); break;
#endif
    case 0: case 1: hipLaunchKernelGGL((attnq_kernel<1, 2, 0, false, 8>), grid, dim3(512), 0, stream, a); break;
    default: PB_CHECK(false, -1, "attention: variant %d is an experiment of the -DPB_DIAG build", variant);
    }
    PB_HIP(hipGetLastError());
    return 0;
}
