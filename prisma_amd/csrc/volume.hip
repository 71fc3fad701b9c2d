// All-pairs correlation volume of the flow band (bands/raft/corr.py:52-60): out[M, N] (fp16) = A[M, 256] . W[N, 256]^T with
// M = N = the 1/8-resolution pixels of a pair (18 360 at 1080p x 0.75) - a GEMM with K = 256 whose 703 MB of output per pair dwarf its
// operands.  On the generic kernels a 128 x 128 tile spends ~5 us in prologue + epilogue around four K tiles of MFMAs
// (tools/volume_gemm_probe.py: 1.7-2.1 TB/s of volume written), so this kernel is A-STATIONARY and PERSISTENT along N:
//   * a workgroup (4 waves) owns 128 source rows, a wave 32 of them: its MFMA fragments for all 16 k-steps live in registers
//     (64 VGPRs), loaded once;
//   * it walks `tiles_per_wg` target tiles of 64 columns; the target features stream through a 4-stage LDS ring (a stage = two K
//     tiles of 64 rows x 128 B = 16 KB, LDS-DMA, same 128-byte rows and XOR swizzle as gemm_kernels.h), three stages in flight, so
//     the next tile's operands arrive under the current tile's MFMAs and stores; every wave reads the whole 64-column tile
//     (2x the LDS fragment traffic per MFMA of a 2 x 2 wave layout - affordable: the matrix pipe is needed a quarter of the time);
//   * the stores are the direct interleaved ones of gemm_kernels.h (a lane owns two adjacent columns, 32 lanes write 128 B of a row).
// Accumulation order over K is that of the generic kernels, so the volume is bit-identical to theirs.
// Waits: loads and stores share vmcnt and return out of order with respect to each other, so a wait that must see a DMA land can only be
// "everything but the newest stage".  A tile's stores are therefore issued one tile late, right after such a wait, and the next one is a
// whole tile away.
// Measured (1080p x 0.75, 31 pairs, 4 levels): 15.4 ms per step on the generic kernels -> 13.1 ms with stores issued at once -> 12.4 ms
// deferred (2.4 TB/s of volume written).  Per tile the matrix pipe and the LDS fragment reads need ~1000 cycles each; the steps (two per
// tile, 512 MFMA cycles) are short against their barrier + wait.  Tried and measured slower: a 3-slot ring at three workgroups per CU
// (13.6 ms); a fifth, load-only producer wave so that the compute waves never wait on their stores (28 ms).
// Round 5, measured slower as well (profiles/r05g_volume_64row_waves.txt): 64 source rows per wave against 32-column tiles - half the LDS
// fragment traffic per MFMA, a lane then owns one column, neighbouring lanes swap a register per row pair and store dwords, 16 lanes
// covering 64 contiguous bytes of a row: bit-identical volume, 12.5 -> 17.6 ms per step.  The kernel is bound by its stores (half-line
// segments cost more than the fragment reads they saved), not by what its K tiles move through the LDS like the GEMM kernels (gemm_n128.h).
#include "common.h"
#include "raft_kernels.h"
#include "../../include/prisma_bands.h"

#define LAUNCH_CHECK() do { PB_HIP(hipGetLastError()); return 0; } while (0)

namespace {

// Bare barriers: __syncthreads() carries a workgroup fence, for which the compiler drains vmcnt - with the LDS-DMA stages and the deferred
// stores in flight that made every step wait for ALL of them (round 3: found in the ISA, `s_waitcnt vmcnt(0) lgkmcnt(0)` in front of the
// second barrier of a tile).  The counted waits below are what orders the DMAs; a wave's fragment reads are consumed by its MFMAs before it
// reaches the next barrier.
#define VOL_BAR()                               \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        asm volatile("s_barrier" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

constexpr int VBM = 128, VBN = 64, VNT = 256, VSTAGES = 4, VSTAGE_BYTES = 2 * VBN * 128;     // a stage = two K tiles of 64 columns

__global__ __launch_bounds__(VNT, 2) void corr_volume_kernel(const f16 *__restrict__ A, int M, const f16 *__restrict__ W, int N, int w_rows,
                                                             f16 *__restrict__ out, int64_t ldo, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;

    // workgroup -> (row tile, group of column tiles); XCD-contiguous remap so that the workgroups sharing a column group share an L2
    const int tilesM = (M + VBM - 1) / VBM, tilesN = (N + VBN - 1) / VBN;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int grp = swz / tilesM, tile_m = swz - grp * tilesM;
    const int m0 = tile_m * VBM + wave * 32;                    // this wave's 32 rows
    const int nt0 = grp * tiles_per_wg;
    const int ntl = (tilesN - nt0) < tiles_per_wg ? (tilesN - nt0) : tiles_per_wg;
    if (ntl <= 0) return;

    // ---- A fragments: the 16 k-steps of this wave's 32 rows, straight from global memory in the MFMA operand layout ----
    f16x8 af[16];
    {
        const int row = m0 + li;
        const f16 *src = A + (int64_t)(row < M ? row : M - 1) * 256 + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) af[ks] = *(const f16x8 *)(src + ks * 16);
    }

    // ---- B staging: stage g = K tiles 2h, 2h + 1 (h = g & 1) of column tile g >> 1, as two 64-row x 128-byte LDS tiles (rows are the
    //      tile's columns in the interleaved order of gemm_kernels.h col_map, 16-byte chunks XOR-swizzled by (row >> 1) & 7);
    //      rows past w_rows read zeros (buffer range check) ----
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(W, (unsigned)((int64_t)w_rows * 512));
    const int srow = tid >> 3, cg = (tid & 7) ^ ((tid >> 4) & 7);
    int b_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = srow + i * 32;
        b_voff[i] = (2 * (r & 31) + (r >> 5)) * 512 + cg * 16;
    }
    auto stage = [&](int g) {
        const int nt = nt0 + (g >> 1), h = g & 1;
        char *dst = smem + (g & (VSTAGES - 1)) * VSTAGE_BYTES + wave * 1024;
        const int soff = nt * (VBN * 512) + h * 256;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16_buf(rsW, b_voff[i], soff + kk * 128, dst + kk * (VBN * 128) + i * 4096);
    };

    const int fsw = (li >> 1) & 7;
    const int b_off[2] = {li * 128, (32 + li) * 128};

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int G = ntl * 2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the A fragments: keeps the DMA counts below exact
    stage(0);
    stage(1);
    if (G > 2) stage(2);
    // af is indexed with compile-time k-steps inside each half (a runtime index would put it in scratch)
#define PB_VOL_HALF(H)                                                                                                  \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                                  \
        const int c = kk * (VBN * 128) + ((2 * ks + lh) ^ fsw) * 16;                                                    \
        const f16x8 b0 = *(const f16x8 *)(sb + b_off[0] + c), b1 = *(const f16x8 *)(sb + b_off[1] + c);                 \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[H * 8 + kk * 4 + ks], b0, acc[0], 0, 0, 0);                  \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[H * 8 + kk * 4 + ks], b1, acc[1], 0, 0, 0);                  \
    }
    // A lane owns columns 2 li, 2 li + 1 of a tile (acc[0], acc[1]) for 16 rows: 32 lanes write 128 B of a row.  A tile's stores are issued
    // one tile LATE (from `pk`), right after the wait of the next tile's first step: the wait that has to count them - loads and stores
    // share vmcnt and return out of order with respect to each other, so a wait that must see a DMA land can only be "everything but the
    // newest stage" - is then a whole tile (two steps) away, and they drain under that tile's MFMAs.
    f16x2 pk[16];
    int pk_n = -1;                                              // first column of the tile held in pk, -1: none
    auto flush = [&]() {
        if (pk_n >= 0 && pk_n < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) *(f16x2 *)(out + (int64_t)m * ldo + pk_n) = pk[r];
            }
        }
    };
    for (int j = 0; j < ntl; ++j) {
        const int g = 2 * j;
        // ---- step g: stages g and g + 1 must have landed; only stage g + 2 (4 DMAs) may stay in flight - and none of the stores ----
        if (g + 2 < G) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VOL_BAR();                                              // ... for every wave; slot (g + 3) & 3 = (g - 1) & 3 is free
        if (g + 3 < G) stage(g + 3);
        flush();                                                // the previous tile's results
        {
            const char *sb = smem + (g & (VSTAGES - 1)) * VSTAGE_BYTES;
            PB_VOL_HALF(0)
        }
        // ---- step g + 1: landed already (waited for above); the barrier frees slot g & 3 for stage g + 4 ----
        VOL_BAR();
        if (g + 4 < G) stage(g + 4);
        {
            const char *sb = smem + ((g + 1) & (VSTAGES - 1)) * VSTAGE_BYTES;
            PB_VOL_HALF(1)
        }
        pk_n = (nt0 + j) * VBN + 2 * li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pk[r] = f16x2{(f16)acc[0][r], (f16)acc[1][r]};
            acc[0][r] = 0.f; acc[1][r] = 0.f;
        }
    }
#undef PB_VOL_HALF
    flush();
}

}  // namespace

int launch_corr_volume(hipStream_t s, const f16 *A, int M, const f16 *W, int N, int w_rows, f16 *out, int64_t ldo) {
    PB_CHECK(A && W && out && M > 0 && N > 0 && N % 8 == 0 && ldo >= N && w_rows >= N, PB_ERR_ARG, "corr_volume: bad arguments");
    static bool once = false;
    const int smem = VSTAGES * VSTAGE_BYTES;
    if (!once) {
        PB_HIP(hipFuncSetAttribute((const void *)corr_volume_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        once = true;
    }
    const int tilesM = (M + VBM - 1) / VBM, tilesN = (N + VBN - 1) / VBN;
    // ~2000 workgroups (4 rounds of 2 per CU) unless the level is small; a workgroup's A load (64 KB) is amortised over its tiles
    int groups = (2048 + tilesM - 1) / tilesM;
    groups = groups < 1 ? 1 : (groups > tilesN ? tilesN : groups);
    const int tpw = (tilesN + groups - 1) / groups;
    groups = (tilesN + tpw - 1) / tpw;
    hipLaunchKernelGGL(corr_volume_kernel, dim3(tilesM * groups), dim3(VNT), smem, s, A, M, W, N, w_rows, out, ldo, tpw);
    LAUNCH_CHECK();
}
