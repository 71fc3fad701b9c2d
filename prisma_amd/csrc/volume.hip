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
// covering 64 contiguous bytes of a row: bit-identical volume, 12.5 -> 17.6 ms per step.
// Round 5, what the parts cost on their own (PB_VOL_ABL, profiles/r05y_volume_ablations.txt; ms per step, the 12.1 ms kernel of that
// visit): stores only 6.2 (4.5 TB/s - tools/probe/volume_store_probe.hip reaches 4.7 with this exact pattern and 5.0 with 512-byte
// runs: the pattern is not the problem), matrix work only 7.4, + DMA 8.3, DMA + stores 9.4, all three 12.1: the parts ADD rather than
// overlap.  Two changes came out of it: buffer stores (flat 64-bit addresses cost a v_mad_u64_u32 + v_lshl_add_u64 per store; and the
// descriptor has to be built from readfirstlane'd halves or every store sits in a waterfall loop) 12.7 -> 11.9 ms, and fragment reads
// issued a quarter tile ahead (see the K loop) 11.9 -> 11.65.  Measured without gain, removed: pad loads so that a tile of stores may
// stay in flight across the wait (16 out-of-range single-lane LDS-DMA loads per tile raise the number of LOADS behind the stage the
// wait is about: 12.4 ms, with 32: 12.9), stores spread two per k-step over the first half of the tile (11.66 = unchanged) or one per
// k-step over the whole tile (12.15), one 8-wave workgroup per CU sharing the target tiles (half the DMA bytes: 13.4 ms, matrix work
// alone 8.6 - the barrier spans twice the waves).
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

template <int ABL>
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
        if (ABL & 4) return;
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
    // Fragment reads run ahead of the MFMAs through two quarter buffers (a quarter = one K tile of 64 = 4 k-steps x 2 column halves = 8
    // ds_read_b128 = 32 registers): the quarters of a tile are q0, q1 (stage g) and q2, q3 (stage g + 1, landed as well when the tile
    // starts); q0 and q1 are requested together, q2 / q3 are requested k-step by k-step into the registers q0 / q1's MFMAs have just
    // consumed.  Written as "two reads, two MFMAs" the compiler kept ONE pair of fragments in flight (118 registers, as if four waves per
    // SIMD were possible - the LDS allows two) and every k-step waited out the LDS latency: the matrix work alone, no DMA and no stores,
    // took 7.4 ms per step against 2.8 ms of MFMA cycles (PB_VOL_ABL, profiles/r05y_volume_ablations.txt).
    // af is indexed with compile-time k-steps (a runtime index would put it in scratch).
    f16x8 qa[8], qb[8];
#define PB_VOL_C(KK, KS) ((KK) * (VBN * 128) + ((2 * (KS) + lh) ^ fsw) * 16)
#define PB_VOL_RD(Q, SB, KK, KS)                                                  \
    Q[2 * (KS)] = *(const f16x8 *)((SB) + b_off[0] + PB_VOL_C(KK, KS));           \
    Q[2 * (KS) + 1] = *(const f16x8 *)((SB) + b_off[1] + PB_VOL_C(KK, KS));
#define PB_VOL_MM(Q, A0, KS)                                                                       \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[(A0) + (KS)], Q[2 * (KS)], acc[0], 0, 0, 0);     \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[(A0) + (KS)], Q[2 * (KS) + 1], acc[1], 0, 0, 0);
#define PB_VOL_PIN() __builtin_amdgcn_sched_barrier(0)
    // A lane owns columns 2 li, 2 li + 1 of a tile (acc[0], acc[1]) for 16 rows: 32 lanes write 128 B of a row.  A tile's stores are issued
    // one tile LATE (from `pk`), right after the wait of the next tile's first step: the wait that has to count them - loads and stores
    // share vmcnt and return out of order with respect to each other, so a wait that must see a DMA land can only be "everything but the
    // newest stage" - is then a whole tile (two steps) away, and they drain under that tile's MFMAs.
    // The stores are buffer stores through a resource that spans exactly this wave's rows (base out + m0 * ldo, range = its valid rows): a
    // 32-bit lane offset + a scalar row offset per store, and rows past M fall outside the range.  Flat 64-bit addresses cost a
    // v_mad_u64_u32 + v_lshl_add_u64 pair per store, which next to a wave that keeps the matrix pipe busy is what
    // tools/probe/store_probe.hip prices at ~770 cycles per access (buffer-addressed: 35).
    f16x2 pk[16];
    int pk_n = -1;                                              // first column of the tile held in pk, -1: none
    const int rows_w = (M - m0) < 32 ? (M - m0 < 0 ? 0 : M - m0) : 32;
    const int ldb = (int)ldo * 2;                               // row stride in bytes (launch_corr_volume checks 32 rows fit 31 bits)
    // (built from readfirstlane'd halves: left to itself the compiler keeps the descriptor in VGPRs and wraps every store in a waterfall loop)
    const uint64_t ob = (uint64_t)(out + (int64_t)m0 * ldo);
    const uint64_t ob_u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(ob >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)ob);
    const __amdgpu_buffer_rsrc_t rsO = make_rsrc((const void *)ob_u, (unsigned)__builtin_amdgcn_readfirstlane(rows_w * ldb));
    const int vo_lane = 4 * lh * ldb + 4 * li;
    // Lanes without a tile to store (none yet, columns past N) carry an offset outside the resource: the store is dropped by the range check.
    // The row of a store travels in the SCALAR offset, which the compiler's builtin documents as excluded from bounds checking: the last
    // row tile (fewer than 32 valid rows) therefore marks its rows past M in the lane offset as well.  (Measured: gfx950 drops such stores
    // on its own - tests/test_gpu_ops.py::test_corr_volume_kernel_values_and_untouched_memory passes without the mask - but nothing
    // guarantees it, and the mask costs a wave-uniform branch.)
    constexpr int VO_NONE = 0x7ffff000;
    int vo_st = VO_NONE;
    const bool rows_full = __builtin_amdgcn_readfirstlane(rows_w) == 32;
    auto flush = [&]() {
        if (ABL & 1) return;
        if (rows_full) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk[r]), rsO, vo_st, ((r & 3) + 8 * (r >> 2)) * ldb, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pk[r]), rsO, row + 4 * lh < rows_w ? vo_st : VO_NONE, row * ldb, 0);
            }
        }
    };
    unsigned keep = 0;                                          // (PB_VOL_ABL: keeps the matrix work alive when the stores are ablated)
    for (int j = 0; j < ntl; ++j) {
        const int g = 2 * j;
        // ---- step g: stages g and g + 1 must have landed; only stage g + 2 (4 DMAs) may stay in flight - and none of the stores ----
        if (g + 2 < G) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        VOL_BAR();                                              // ... for every wave; slot (g + 3) & 3 = (g - 1) & 3 is free
        if (g + 3 < G) stage(g + 3);
        flush();                                                // the previous tile's results
        const char *sb0 = smem + (g & (VSTAGES - 1)) * VSTAGE_BYTES, *sb1 = smem + ((g + 1) & (VSTAGES - 1)) * VSTAGE_BYTES;
        if (!(ABL & 2)) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { PB_VOL_RD(qa, sb0, 0, ks) }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { PB_VOL_RD(qb, sb0, 1, ks) }
            PB_VOL_PIN();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { PB_VOL_MM(qa, 0, ks) PB_VOL_RD(qa, sb1, 0, ks) PB_VOL_PIN(); }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { PB_VOL_MM(qb, 4, ks) PB_VOL_RD(qb, sb1, 1, ks) PB_VOL_PIN(); }
        }
        // ---- every read of stage g has been consumed: the barrier frees slot g & 3 for stage g + 4 ----
        VOL_BAR();
        if (g + 4 < G) stage(g + 4);
        if (!(ABL & 2)) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { PB_VOL_MM(qa, 8, ks) }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { PB_VOL_MM(qb, 12, ks) }
        }
        pk_n = (nt0 + j) * VBN + 2 * li;
        vo_st = pk_n < N ? vo_lane + (nt0 + j) * (VBN * 2) : VO_NONE;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pk[r] = f16x2{(f16)acc[0][r], (f16)acc[1][r]};
            acc[0][r] = 0.f; acc[1][r] = 0.f;
            if (ABL & 1) keep ^= __builtin_bit_cast(unsigned, pk[r]);
        }
    }
#undef PB_VOL_C
#undef PB_VOL_RD
#undef PB_VOL_MM
#undef PB_VOL_PIN
    flush();
    if ((ABL & 1) && keep == 0x12345679u) out[0] = (f16)1.f;
}

}  // namespace

int launch_corr_volume(hipStream_t s, const f16 *A, int M, const f16 *W, int N, int w_rows, f16 *out, int64_t ldo) {
    PB_CHECK(A && W && out && M > 0 && N > 0 && N % 8 == 0 && ldo >= N && w_rows >= N && ldo < (1 << 24), PB_ERR_ARG, "corr_volume: bad arguments");
    static bool once = false;
    const int smem = VSTAGES * VSTAGE_BYTES;
    static int abl = 0;                                         // PB_VOL_ABL (-DPB_DIAG builds only; wrong results): 1 no stores, 2 no fragment reads / MFMAs, 4 no DMA
    if (!once) {
        PB_HIP(hipFuncSetAttribute((const void *)corr_volume_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
#ifdef PB_DIAG
        abl = pb_env_int("PB_VOL_ABL", 0);
        PB_HIP(hipFuncSetAttribute((const void *)corr_volume_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        PB_HIP(hipFuncSetAttribute((const void *)corr_volume_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        PB_HIP(hipFuncSetAttribute((const void *)corr_volume_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        PB_HIP(hipFuncSetAttribute((const void *)corr_volume_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        PB_HIP(hipFuncSetAttribute((const void *)corr_volume_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
#endif
        once = true;
    }
    const int tilesM = (M + VBM - 1) / VBM, tilesN = (N + VBN - 1) / VBN;
    // ~2000 workgroups (4 rounds of 2 per CU) unless the level is small; a workgroup's A load (64 KB) is amortised over its tiles
    int groups = (2048 + tilesM - 1) / tilesM;
    groups = groups < 1 ? 1 : (groups > tilesN ? tilesN : groups);
    const int tpw = (tilesN + groups - 1) / groups;
    groups = (tilesN + tpw - 1) / tpw;
#define PB_VOL_LAUNCH(X) hipLaunchKernelGGL(corr_volume_kernel<X>, dim3(tilesM * groups), dim3(VNT), smem, s, A, M, W, N, w_rows, out, ldo, tpw)
    switch (abl) {
#ifdef PB_DIAG
        case 1: PB_VOL_LAUNCH(1); break;
        case 2: PB_VOL_LAUNCH(2); break;
        case 4: PB_VOL_LAUNCH(4); break;
        case 5: PB_VOL_LAUNCH(5); break;
        case 6: PB_VOL_LAUNCH(6); break;
#endif
        default: PB_VOL_LAUNCH(0);
    }
#undef PB_VOL_LAUNCH
    LAUNCH_CHECK();
}
