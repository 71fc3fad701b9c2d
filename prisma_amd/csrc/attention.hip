// Fused softmax(Q K^T) V for the DINOv2 blocks (reference dinov2/layers/attention.py:49-62:
// q * 64^-0.5, q @ k^T, softmax, @ v) on gfx950 - flash style, the N x N score matrix never
// leaves registers.  head_dim is 64 for every encoder size.
//
// Layouts (written by the qkv GEMM epilogue, gemm.hip EPI_QKV):
//   Q, K : [b, head, ntp, 64] fp16, Q already multiplied by 64^-0.5 (exact in fp16)
//   Vt   : [b, head, 64, ntp] fp16 (V transposed: keys contiguous)
//   O    : [b * ntp, heads * 64] fp16 row-major (the proj GEMM's A operand)
//
// Workgroup = 4 waves = 128 query rows of one (b, head); each wave owns 32 query rows.
// Per 64-key tile:
//   S^T[key][q] = K Q^T  with v_mfma_f32_32x32x16_f16 (A = K rows, B = Q rows): a lane then holds
//                 32 scores of ONE query column, so the row max / row sum are in-lane plus one
//                 exchange with lane ^ 32.
//   The K rows of a 32-key sub-tile are fed in the order swap_bits23(row): with that order
//   accumulator registers 8j..8j+7 of a lane are 8 CONSECUTIVE keys, i.e. exactly the B-operand
//   fragment of the second matmul - P never moves between lanes.
//   O^T[d][q] += Vt[d][key] P^T[key][q]  (A = Vt rows, one ds_read_b128 per fragment).
// K and Vt tiles go HBM -> LDS with global_load_lds_dwordx4, double buffered, XOR-swizzled
// through the source address exactly like gemm.hip.  Keys >= ntok are masked to -inf.
#include "common.h"

#include <stdlib.h>

namespace {

struct AttnArgs {
    const f16 *q, *k, *vt;
    f16 *o;
    int ntp, ntok, heads, ldo, nq, nb;
};

template <int OCC>
__global__ __launch_bounds__(256, OCC) void attn_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 16384];   // 2 x (K tile 8 KB + Vt tile 8 KB)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    // All q-blocks of one (b, head) run on ONE XCD (blockIdx % 8) so its K / Vt stream is fetched into one L2 only:
    // with the natural (qblk, head, b) order the 20 q-blocks of a head spread over all 8 XCDs and rocprofv3 showed
    // 1.4 GB of fabric fetches per launch against 0.48 GB of q/k/v.
    const int nq = p.nq, nbh = p.heads * p.nb;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bhi = (slot / nq) * 8 + xcd, qblk = slot % nq;
    if (bhi >= nbh) return;
    const int b = bhi / p.heads, head = bhi - b * p.heads;
    const int64_t bh = bhi;
    const f16 *Q = p.q + bh * p.ntp * 64;
    const f16 *K = p.k + bh * p.ntp * 64;
    const f16 *Vt = p.vt + bh * 64 * p.ntp;

    // ---- Q fragments (B operand of S^T): lane (q = li, half lh) holds Q[q][16*s + 8*lh .. +8] ----
    const int qrow = qblk * 128 + wave * 32 + li;
    const int qrc = qrow < p.ntp ? qrow : p.ntp - 1;
    f16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const f16x8 *)(Q + (int64_t)qrc * 64 + 16 * s + 8 * lh);

    // ---- staging: each thread moves 2 chunks of K and 2 of Vt per tile ----
    //      through the buffer path (cheaper to issue than the flat global_load_lds): per-lane byte offsets are loop
    //      invariant, the tile advance is a scalar offset, reads past the (b, head) block return zeros
    const int srow = tid >> 3;                               // 0..31 (+32 for the second chunk)
    const int cg = (tid & 7) ^ ((tid >> 4) & 7);
    const __amdgpu_buffer_rsrc_t rsK = make_rsrc(K, (unsigned)(p.ntp * 64 * 2)), rsV = make_rsrc(Vt, (unsigned)(64 * p.ntp * 2));
    int kvo[2], vvo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = srow + 32 * i;
        kvo[i] = r * 128 + cg * 16;
        vvo[i] = (r * p.ntp + cg * 8) * 2;
    }
    auto stage = [&](int buf, int t) {
        char *sK = smem + buf * 16384 + wave * 1024;
        char *sV = sK + 8192;
        const int key0 = t * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            glds16_buf(rsK, kvo[i], key0 * 128, sK + i * 4096);
            glds16_buf(rsV, vvo[i], key0 * 2, sV + i * 4096);
        }
    };

    const int fsw = (li >> 1) & 7;
    // K fragment rows: MFMA row i of sub-tile kt reads key 32*kt + swap_bits23(i)
    const int kperm = (li & 19) | ((li & 4) << 1) | ((li & 8) >> 1);
    const int fswk = (kperm >> 1) & 7;

    f32x16 oacc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float mrun = -1e30f, lrun = 0.f;
    constexpr float L2E = 1.4426950408889634f;

    const int nt = (p.ntok + 63) >> 6;
    stage(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stage((t + 1) & 1, t + 1 < nt ? t + 1 : t);            // past the end: re-read the last tile into the idle buffer
        const char *sK = smem + (t & 1) * 16384;
        const char *sV = sK + 8192;

        // ---- S^T = K Q^T : two 32-key sub-tiles ----
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 kf = *(const f16x8 *)(sK + (kt * 32 + kperm) * 128 + (((2 * ks + lh) ^ fswk) * 16));
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kt], 0, 0, 0);
            }
        }
        // register g of sub-tile kt in lane (q, lh) is key  t*64 + 32*kt + 16*(g>>3) + 8*lh + (g&7)
        if (t == nt - 1) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int key = t * 64 + 32 * kt + 16 * (g >> 3) + 8 * lh + (g & 7);
                    if (key >= p.ntok) s[kt][g] = -1e30f;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 16; ++g) mx = fmaxf(mx, s[kt][g]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // Deferred rescale: keep the old reference max while no row max grew by more than 4 (P <= e^4 = 55, still
        // exact to fp16's 11 bits; l and O accumulate in fp32), so most tiles skip the O rescale entirely.
        if (__any(mx > mrun + 4.0f)) {
            const float mnew = fmaxf(mrun, mx);
            const float alpha = exp2f((mrun - mnew) * L2E);
            mrun = mnew;
            lrun *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
        }
        const float ml = mrun * L2E;                         // p = 2^(s*log2e - m*log2e): one fma + one v_exp
        float psum = 0.f;
        f16x8 pf[4];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][g], L2E, -ml));
                psum += e;
                pf[kt * 2 + (g >> 3)][g & 7] = (f16)e;
            }
        lrun += psum;

        // ---- O^T += Vt P^T : 4 k-steps of 16 keys, 2 d-tiles ----
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const f16x8 vf = *(const f16x8 *)(sV + (d * 32 + li) * 128 + (((2 * j + lh) ^ fsw) * 16));
                oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[j], oacc[d], 0, 0, 0);
            }
    }

    const float ltot = lrun + __shfl_xor(lrun, 32);
    const float inv = 1.f / ltot;
    if (qrow < p.ntok) {
        f16 *orow = p.o + ((int64_t)b * p.ntp + qrow) * p.ldo + head * 64;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 r;
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] = (f16)(oacc[d][g * 4 + j] * inv);
                *(f16x4 *)(orow + d * 32 + 8 * g + 4 * lh) = r;
            }
    }
}

}  // namespace

int launch_attention(hipStream_t stream, const f16 *q, const f16 *k, const f16 *vt, f16 *o, int B, int heads,
                     int ntp, int ntok, int ldo) {
    const int nq = (ntok + 127) / 128;
    AttnArgs a{q, k, vt, o, ntp, ntok, heads, ldo, nq, B};
    dim3 grid(8 * nq * ((B * heads + 7) / 8));
    static int variant = -1;
    if (variant < 0) { const char *e = getenv("PB_ATTN_OCC"); variant = e ? atoi(e) : 2; }
    if (variant == 3) hipLaunchKernelGGL(attn_kernel<3>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(attn_kernel<2>, grid, dim3(256), 0, stream, a);
    PB_HIP(hipGetLastError());
    return 0;
}
