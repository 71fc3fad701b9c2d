// 384 x 128 x 64 "ping-pong" GEMM / implicit-GEMM convolution for the layers with N <= 128 (round 5).
//
// Why it exists.  The generic 128 x 128 tile (gemm_kernel: four waves of 64 x 64, two workgroups per CU) keeps the matrix pipe busy 41 % of
// the time on the RAFT update block's and encoders' N <= 128 convolutions and on the DPT head's output_conv1: 43 ms of a 258 ms step.
// What a ping-pong K loop costs (profiles/r05b_dma_probe.txt, r05m_lds_read_probe.txt): its K tile's fragment reads at the
// LDS read ports' 256 bytes per cycle and CU PLUS its LDS-DMA pieces at ~41 (one 1 KB piece per ~25 cycles and CU while the loading wave
// group issues them) - 3.9 cycles per KB read + 24.6 per KB staged predicts gemm8_kernel's K tile to a cycle (2323 / 2324 against 2048
// MFMA-bound).  Neither the L2 -> LDS transport (59.5 bytes per cycle and CU when alone) nor LDS bandwidth as such is the bound.  A first
// ping-pong kernel of 256 x 128 with 64 x 64 wave tiles (128 KB read + 48 KB staged per K tile: 1680 predicted, 1764-1869 measured against
// 1024 MFMA-bound, profiles/r05a_n128_tile_stamps.txt) confirmed it: the staggered wave groups remove the issue stalls of the generic tile,
// and the loading group's reads + DMA issue then take 1.75x the partner's MFMA segment.
//
// This kernel: BN = 128 with a 96 x 64 wave tile (3 x 2 MFMA tiles, 96 accumulator registers), i.e. 0.83 KB of fragment reads and 0.33 KB of
// DMA per MFMA (160 + 64 KB per K tile: 2198 cycles predicted, 2200-2370 measured against 1536 MFMA-bound - the DMA term alone is 1574, which
// is why ~70 % is what BN = 128 can reach), in gemm8_kernel's schedule:
//   * 8 waves = 2 wave groups (rows 0-191 / 192-383) staggered by one barrier, 2 x 2 waves of 96 x 64 each;
//   * a K tile is three phases of 8 MFMAs per wave, one per 32-row tile rt:
//         p0: read B(j0)[4] B(j1)[4] A(0)[4] | mfma (0, *)      p1: read A(1)[4] | mfma (1, *)      p2: read A(2)[4] | mfma (2, *)
//   * LDS: two K-tile buffers x {A 48 KB, B 16 KB} (+ the conv K-walk table); a K tile is staged as eight 8 KB slots, one DMA per thread
//     each - A[g][rt] = the 64 rows wave group g reads in phase rt, B[j] = the B rows of column tile j - in consumption order:
//         p0(t): A[.][1] B[1] of tile t + 1      p1(t): A[.][2] of tile t + 1      p2(t): A[.][0] B[0] of tile t + 2
//     behind one counted wait per phase that leaves the two newest phases' DMAs in flight (vmcnt 6 / 5 / 5), and bare barriers.
//     RAW: a slot staged in phase P is certified by both groups' waits of phase P + 2 and first read in phase >= P + 3.
//     WAR: a slot is re-staged two phases (four barrier segments) after the phase that read it.
//
// PERSISTENT workgroups, one per CU, walk the tiles vb = blockIdx.x + k gridDim.x of the XCD-contiguous order.  The first version ran one
// workgroup per tile and lost to the generic tile although its K loop is faster (profiles/r05c_n128_384_tile_stamps.txt: 2200 cycles per
// 384 x 128 x 64 against ~2 x 1300 for the same work): a tile spent 10-12.7 k cycles between launch and its first MFMA - the K-walk table, two
// integer divisions per staging slot, eleven DMAs and their way from the HBM with every CU asking at once - and 2.7 k in the epilogue, around
// a 40-85 k-cycle K loop; the generic tile hides the same costs behind its second workgroup per CU.  Here the table is built once, and the next
// tile's addresses and first eleven DMAs are issued BEFORE the epilogue (the staging buffers are dead by then; EPI_STD tiles that store
// straight from the accumulators touch no LDS), so they land while the stores are issued; one `s_waitcnt vmcnt(0)` in front of the next
// K loop then covers them and the stores (PB_G8N_PERSIST=0: one workgroup per tile, for A/B runs).
// Operands, swizzle, K walk, MX-fp8 tiles and epilogues are gemm8_kernel's / gemm_kernel's (gemm_kernels.h), so every EPI_STD launch kind of
// the bands runs unchanged.
#pragma once
#include "gemm_kernels.h"

namespace {

constexpr int G8N_BM = 384;

template <int AMODE, int EPI, bool BUFP, bool MX>
__global__ __launch_bounds__(512) void gemm8n_kernel(const GemmArgs p) {
    constexpr int BM = G8N_BM, BN = 128;
    constexpr int BUF = 65536, BOFF = 49152, NBUF = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wr = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;        // wave group, 96-row block inside the group, 64-column block
    long long ts0 = 0, ts1 = 0, ts2 = 0, tr0 = 0;
    if (p.dbg) { ts0 = __builtin_readcyclecounter(); tr0 = wall_clock64(); }

    // ---- tiles of the launch (the grid may be smaller: persistent workgroups), XCD-contiguous remap (bijective for any tile count;
    //      workgroup b runs on XCD b % 8 and the persistent stride is a multiple of 8) ----
    const int tilesN = (p.N + BN - 1) / BN;
    const int nwg = p.ntiles > 0 ? p.ntiles : (int)gridDim.x;
    const int qd = nwg >> 3, rm = nwg & 7;
    int m0 = 0, n0 = 0, swz = 0;

    // ---- staging geometry.  LDS A rows are ordered by slot: row (g * 3 + rt) * 64 + wm * 32 + i holds tile row g * 192 + wm * 96 + rt * 32 + i;
    //      this wave's DMA of slot (g, rt) covers LDS rows (g * 3 + rt) * 64 + 8 wave .. + 7.  B slot j covers the LDS rows
    //      (wave >> 2) * 64 + 32 j + (wave & 3) * 8 .. + 7 (column tile j of wave column wave >> 2).  Every row0 is a multiple of 8 with
    //      (row0 >> 3) & 1 == wave & 1, so ONE swizzled source chunk serves all eight slots of a lane ----
    const int cld = p.cLd ? p.cLd : p.cC, padx = p.cPadX >= 0 ? p.cPadX : p.cPad;
    const int lrow = lane >> 3;
    const int cgw = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const int nk = p.K >> 6;
    const f16 *a_ptr[2][3], *b_ptr[2];
    int a_yx[2][3], a_pix0[2][3];
    unsigned a_voff[2][3], b_voff[2];
    __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, 0u), rsW = make_rsrc(p.W, 0u);
    if constexpr (BUFP) {
        rsW = make_rsrc(p.W, (unsigned)((int64_t)((p.N + 255) / 256 * 256) * p.K * 2));
        if constexpr (AMODE == A_DENSE) rsA = make_rsrc(p.A, (unsigned)((int64_t)p.M * p.lda * 2));
    }
    // tile vb of the launch -> (m0, n0) and this thread's staging addresses
    auto setup = [&](int vb) {
    const int xcd = vb & 7;
    swz = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (vb >> 3);
    const int tile_m = swz / tilesN, tile_n = swz - tile_m * tilesN;
    m0 = tile_m * BM; n0 = tile_n * BN;
    int bimg0 = 0;
    if constexpr (BUFP && AMODE == A_CONV) {
        const int ohw = p.cOH * p.cOW, nimg = p.M / ohw;
        const int64_t img = (int64_t)p.cH * p.cW * cld;
        bimg0 = p.bufmode == 2 ? m0 / ohw : 0;
        rsA = make_rsrc(p.A + bimg0 * img, (unsigned)((p.bufmode == 2 ? (nimg - bimg0 < 2 ? nimg - bimg0 : 2) : nimg) * img * 2));
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) {
            const int m = m0 + g * 192 + (wave >> 2) * 96 + rt * 32 + (wave & 3) * 8 + lrow;
            if constexpr (AMODE == A_DENSE) {
                const int mc = m < p.M ? m : p.M - 1;
                a_ptr[g][rt] = p.A + (int64_t)mc * p.lda + cgw * 8;
                a_yx[g][rt] = a_pix0[g][rt] = 0;
                a_voff[g][rt] = (unsigned)((a_ptr[g][rt] - p.A) * 2);
            } else {
                const int ohw = p.cOH * p.cOW;
                const int b = m / ohw, rem = m - b * ohw;
                const int oy = rem / p.cOW, ox = rem - oy * p.cOW;
                a_ptr[g][rt] = p.A + (int64_t)b * p.cH * p.cW * cld + cgw * 8;
                const int iy0 = oy * p.cStride - p.cPad, ix0 = ox * p.cStride - padx;
                a_yx[g][rt] = (int)(m < p.M ? tap_mask(iy0, ix0, p.cH, p.cW) : 0u);
                a_pix0[g][rt] = (iy0 * p.cW + ix0) * cld;
                // byte offset of tap (0, 0) of this lane's pixel inside the resource (the pixel part may be "negative": only in-range taps use the sum)
                a_voff[g][rt] = (unsigned)((int64_t)(b - bimg0) * p.cH * p.cW * cld * 2) + cgw * 16 + (unsigned)(a_pix0[g][rt] * 2);
            }
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave >> 2) * 64 + 32 * j + (wave & 3) * 8 + lrow;
        b_ptr[j] = p.W + (int64_t)(n0 + col_map(r, epi_interleaved<EPI, 2>())) * p.K + cgw * 8;
        b_voff[j] = (unsigned)((b_ptr[j] - p.W) * 2);
    }
    };

    const unsigned *ktab = (const unsigned *)(smem + NBUF * BUF);
    unsigned e_nxt = 0, e_cur = 0;
    // the two A slots (g = 0, 1) of row tile rt of K tile kt_ (tiles past the end re-read the last one into a slot nobody reads any more: the
    // loop stays branch free and every phase uses the same counted wait).  Calls come in the order rt 0, 1, 2 of tile k, rt 0 of tile k + 1, ...:
    // the rt = 0 call advances the K-walk cursor
    auto stage_a = [&](int kt_, int rt) {
        const int ktc = kt_ < nk ? kt_ : nk - 1;
        const int kt = (p.kwrap && ktc >= p.kwrap) ? ktc - p.kwrap : ktc;         // split-fp16 segments re-read A (gemm.h)
        char *base = smem + (kt_ & 1) * BUF + (rt * 64 + 8 * wave) * 128;
        unsigned tapoff2 = 0, tapsel = 0;
        if constexpr (AMODE == A_CONV) {
            if (rt == 0) {
                e_cur = (unsigned)__builtin_amdgcn_readfirstlane((int)e_nxt);
                e_nxt = ktab[ktc + 1 < nk ? ktc + 1 : nk - 1];
            }
            tapoff2 = ktab_bytes(e_cur); tapsel = ktab_sel(e_cur);
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            char *dst = base + g * (3 * 64 * 128);
            if constexpr (AMODE == A_DENSE) {
                if constexpr (BUFP) glds16_buf(rsA, (int)a_voff[g][rt], kt * 128, dst);
                else glds16(a_ptr[g][rt] + kt * 64, dst);
            } else {
                const bool ok = ((unsigned)a_yx[g][rt] & tapsel) == tapsel;        // tap_mask: an AND and a compare per DMA
                if constexpr (BUFP) glds16_buf(rsA, (int)(ok ? a_voff[g][rt] + tapoff2 : 0xFFFFFF00u), 0, dst);      // out-of-range offsets read zeros
                else glds16(ok ? a_ptr[g][rt] + (a_pix0[g][rt] + (int)(tapoff2 >> 1)) : p.zero, dst);
            }
        }
    };
    auto stage_b = [&](int kt_, int j) {
        const int kt = kt_ < nk ? kt_ : nk - 1;
        char *dst = smem + (kt_ & 1) * BUF + BOFF + ((wave >> 2) * 64 + 32 * j + (wave & 3) * 8) * 128;
        if constexpr (BUFP) glds16_buf(rsW, (int)b_voff[j], kt * 128, dst);
        else glds16(b_ptr[j] + kt * 64, dst);
    };

    // ---- fragment addressing (gemm8_kernel's): chunk(ks) = (lh ^ fsw) ^ 2 ks  ->  byte offset = c0 ^ (32 ks) ----
    const int li = lane & 31, lh = lane >> 5;
    const int c0 = (lh ^ ((li >> 1) & 7)) * 16;
    const int a_base = (wr * 192 + wm * 32 + li) * 128;     // + rt * 8192
    const int b_base = BOFF + (wn * 64 + li) * 128;         // + j * 4096

    f32x16 acc[3][2];
    f16x8 fa[4], fb0[4], fb1[4];
    const int mxa = p.mx_scale_a * 0x01010101, mxb = p.mx_scale_b * 0x01010101;

    if constexpr (AMODE == A_CONV) {                     // the K walk of the launch (conv_ktab_entry), once per workgroup
        for (int t = tid; t < nk; t += 512) ((unsigned *)(smem + NBUF * BUF))[t] = conv_ktab_entry(p, cld, t);
        __syncthreads();
    }
    // a tile's prologue: what the phases -3, -2, -1 of the steady state would have staged, behind K tile 0's first slots (eleven DMAs per thread)
    auto prologue = [&]() {
        if constexpr (AMODE == A_CONV) e_nxt = ktab[0];
        stage_a(0, 0); stage_b(0, 0);
        stage_a(0, 1); stage_b(0, 1);
        stage_a(0, 2);
        stage_a(1, 0); stage_b(1, 0);
    };

    // the 8 MFMAs of one phase: row tile RT against both column tiles
    auto mm = [&](auto fp8_tag, auto rt_tag) {
        constexpr bool FP8 = decltype(fp8_tag)::value;
        constexpr int RT = decltype(rt_tag)::value;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (FP8) {
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                acc[RT][0] = mfma_mx8(fa[2 * q2], fa[2 * q2 + 1], fb0[2 * q2], fb0[2 * q2 + 1], acc[RT][0], mxa, mxb);
                acc[RT][1] = mfma_mx8(fa[2 * q2], fa[2 * q2 + 1], fb1[2 * q2], fb1[2 * q2 + 1], acc[RT][1], mxa, mxb);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                acc[RT][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks], fb0[ks], acc[RT][0], 0, 0, 0);
                acc[RT][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks], fb1[ks], acc[RT][1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    auto tile = [&](auto fp8_tag, int t) {
        const char *sb = smem + (t & 1) * BUF;
        // ================= p0 =================
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb0[ks] = *(const f16x8 *)(sb + b_base + (c0 ^ (ks * 32)));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[ks] = *(const f16x8 *)(sb + a_base + (c0 ^ (ks * 32)));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb1[ks] = *(const f16x8 *)(sb + b_base + 4096 + (c0 ^ (ks * 32)));
        stage_a(t + 1, 1); stage_b(t + 1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        PB_BAR();
        mm(fp8_tag, std::integral_constant<int, 0>{});
        PB_BAR();
        // ================= p1 =================
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[ks] = *(const f16x8 *)(sb + a_base + 8192 + (c0 ^ (ks * 32)));
        stage_a(t + 1, 2);
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        PB_BAR();
        mm(fp8_tag, std::integral_constant<int, 1>{});
        PB_BAR();
        // ================= p2 =================
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[ks] = *(const f16x8 *)(sb + a_base + 16384 + (c0 ^ (ks * 32)));
        stage_a(t + 2, 0); stage_b(t + 2, 0);
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        PB_BAR();
        mm(fp8_tag, std::integral_constant<int, 2>{});
        PB_BAR();
    };

    int vb = blockIdx.x;
    setup(vb);
    prologue();
    bool pf = false;                                         // this tile's prologue was issued in front of the previous tile's epilogue
    while (true) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (pf) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the eleven DMAs and, behind them, the previous tile's stores
    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");        // A[.][0], B[0], A[.][1], B[1] of K tile 0 landed (this wave's share)
    PB_BAR();
    if (wr == 1) PB_BAR();                                   // stagger the second wave group by one barrier
    if (p.dbg) ts1 = __builtin_readcyclecounter();
    if constexpr (!MX) {
        for (int t = 0; t < nk; ++t) tile(std::false_type{}, t);
    } else {                                         // per period (mx_period tiles; 0 = the whole K axis): fp16 tiles, then fp8 tiles
        const int per = p.mx_period > 0 ? p.mx_period : nk;
        const int n16 = p.nk16 > 0 && p.nk16 < per ? p.nk16 : per;
        for (int t0 = 0; t0 < nk; t0 += per) {
            const int e16 = t0 + n16 < nk ? t0 + n16 : nk, e8 = t0 + per < nk ? t0 + per : nk;
            for (int t = t0; t < e16; ++t) tile(std::false_type{}, t);
            for (int t = e16; t < e8; ++t) tile(std::true_type{}, t);
        }
    }
    if (wr == 0) PB_BAR();                                   // re-align the two wave groups
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                         // staging buffers are dead; the epilogue may use the LDS
    if (p.dbg) ts2 = __builtin_readcyclecounter();

    // ---- this tile's epilogue, with the next tile's addresses and first DMAs in front of it where the epilogue touches no LDS: interior tiles of
    //      the direct fp16 epilogue (run_epilogue: EPI_STD with interleaved columns stores straight from the accumulators) ----
    const int em0 = m0, en0 = n0, eswz = swz;
    vb += gridDim.x;
    const bool more = vb < nwg;
    pf = more && p.prefetch && EPI == EPI_STD && em0 + BM <= p.M && en0 + BN <= p.N;
    if (pf) { setup(vb); prologue(); }
    run_epilogue<EPI, 3, 2, MX>(p, acc, smem, wave, lane, em0 + wr * 192 + wm * 96, en0 + wn * 64, en0);
    if (p.dbg && tid == 0) {                                 // per TILE, same record as gemm8_kernel's (tools/gemm_stamps.py)
        long long *d = p.dbg + (long long)eswz * 8;
        const long long t_issue = __builtin_readcyclecounter();
        d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = t_issue; d[4] = tr0; d[5] = wall_clock64();
        d[6] = t_issue; d[7] = eswz;
        ts0 = t_issue; tr0 = d[5];
    }
    if (!more) break;
    if (!pf) {
        __syncthreads();                                     // the LDS patches of this tile's epilogue are dead
        setup(vb);
        prologue();
    }
    }
}

template <int AMODE, int EPI, bool MX, bool BUFP>
int launch_g8n_impl(hipStream_t stream, const GemmArgs &a) {
    constexpr int SMEM = 2 * 65536 + (AMODE == A_CONV ? KTAB_BYTES : 0);
    auto kern = gemm8n_kernel<AMODE, EPI, BUFP, MX>;
    static char name[96];
    if (!name[0]) snprintf(name, sizeof(name), "gemm8n_kernel<%d, %d, %s, %s>", AMODE, EPI, BUFP ? "true" : "false", MX ? "true" : "false");
    pb_gemm_set_last_kernel(name);
    static bool attr_set = false;
    if (!attr_set) {
        PB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set = true;
    }
    const int tilesM = (a.M + G8N_BM - 1) / G8N_BM, tilesN = (a.N + 127) / 128;
    static int ncu = 0, persist = 1;
    if (!ncu) {
        int dev = 0;
        PB_HIP(hipGetDevice(&dev));
        PB_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        persist = pb_env_int("PB_G8N_PERSIST", 1);
    }
    GemmArgs b = a;
    b.ntiles = tilesM * tilesN;
    static const int prefetch = pb_env_int("PB_G8N_PREFETCH", 1);
    b.prefetch = prefetch;
    const int grid = persist && b.ntiles > ncu ? ncu : b.ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SMEM, stream, b);
    PB_HIP(hipGetLastError());
    return 0;
}

template <int AMODE, int EPI, bool MX>
int launch_g8n(hipStream_t stream, const GemmArgs &a) {
    GemmArgs b = a;
    b.bufmode = buffer_mode(AMODE, a, G8N_BM);
    if (b.bufmode) return launch_g8n_impl<AMODE, EPI, MX, true>(stream, b);
    return launch_g8n_impl<AMODE, EPI, MX, false>(stream, a);
}

}  // namespace
