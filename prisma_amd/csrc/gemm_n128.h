// 256 x 128 x 64 "ping-pong" GEMM / implicit-GEMM convolution for the layers with N <= 128 (round 5).
//
// Why it exists (profiles/r04y_update_block_sq_counters.txt): the generic 128 x 128 tile (gemm_kernel, four waves of
// 64 x 64, two workgroups per CU) holds the matrix pipe busy 41 % of the time on the RAFT update block's and encoders' N <= 128
// convolutions and on the DPT head's output_conv1 - 43 ms of a 258 ms step at 0.15-0.17 of peak.  Its waves all issue their LDS-DMA pieces
// (8 per 16 MFMAs, 60-185 issue cycles each) and then all run their MFMAs: nothing makes the two waves of a SIMD take turns, and a deeper
// prefetch (round 3's 8-wave 256 x 128 tile on three stages) did not help because the launches do not wait for DMA latency - they wait
// for DMA ISSUE.  What does make them take turns is gemm8_kernel's schedule: two wave groups staggered by one barrier, one loading
// while the other multiplies.  This kernel is that schedule for BN = 128:
//
//   * 8 waves = 2 wave groups (rows 0-127 / 128-255) x 4 waves of 64 x 64 (2 x 2 MFMA tiles, 64 accumulator registers);
//   * a K tile is two phases of 8 MFMAs per wave:   p0: read A[8] + B(j0)[4] | mfma (*, j0)      p1: read B(j1)[4] | mfma (*, j1)
//   * LDS: THREE K-tile buffers x {A 32 KB, B 16 KB} = 144 KB (+ the conv K-walk table); a K tile is staged as six 8 KB slots, one DMA
//     per thread each - A_0..A_3 = A rows {64 s ..}, B_0 / B_1 = the B rows of column tiles j0 / j1 of both wave columns - three slots
//     per phase, two K tiles ahead:        p0(t): A_0 A_1 A_2 of tile t + 2          p1(t): A_3 B_0 B_1 of tile t + 2
//     behind ONE counted wait per phase, `s_waitcnt vmcnt(6)` (two phases' DMAs stay in flight), and bare barriers.
//     RAW: a slot staged in phase P is certified by both groups' waits of phase P + 2 and first read in phase >= P + 3.
//     WAR: the buffer of tile t + 2 = the buffer of tile t - 1, whose last fragment read (B_1, phase 2 t - 1) is consumed by MFMAs two
//     segments before the first DMA into it is issued.
//   * 6 DMA pieces and 16 ds_read_b128 per 16 MFMAs and wave (the generic tile: 8 and 16; gemm8_kernel: 8 and 24 per 32).
//
// One workgroup per tile (no persistent loop: it bought gemm8_kernel 0.5 %); operands, swizzle, K walk, MX-fp8 tiles and epilogues are
// gemm8_kernel's / gemm_kernel's (gemm_kernels.h), so every EPI_STD launch kind of the bands runs unchanged.
#pragma once
#include "gemm_kernels.h"

namespace {

template <int AMODE, int EPI, bool BUFP, bool MX>
__global__ __launch_bounds__(512) void gemm8n_kernel(const GemmArgs p) {
    constexpr int BM = 256, BN = 128;
    constexpr int BUF = 49152, BOFF = 32768, NBUF = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wr = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;        // wave group, 64-row block inside the group, 64-column block
    long long ts0 = 0, ts1 = 0, ts2 = 0, tr0 = 0;
    if (p.dbg) { ts0 = __builtin_readcyclecounter(); tr0 = wall_clock64(); }

    // ---- tile id with XCD-contiguous remap (bijective for any grid size) ----
    const int tilesN = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int tile_m = swz / tilesN, tile_n = swz - tile_m * tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- staging geometry: slot s of A covers rows 64 s .. 64 s + 63, this wave's DMA rows 64 s + 8 wave .. + 7; slot j of B covers the
    //      LDS rows (wave >> 2) * 64 + 32 j + (wave & 3) * 8 .. + 7 (column tile j of wave column wave >> 2).  Every row0 is a multiple of 8
    //      with (row0 >> 3) & 1 == wave & 1, so ONE swizzled source chunk serves all six slots of a lane ----
    const int cld = p.cLd ? p.cLd : p.cC, padx = p.cPadX >= 0 ? p.cPadX : p.cPad;
    const int lrow = lane >> 3;
    const int cgw = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const int nk = p.K >> 6;
    const f16 *a_ptr[4], *b_ptr[2];
    int a_yx[4], a_pix0[4];
    unsigned a_voff[4], b_voff[2];
    __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, 0u), rsW = make_rsrc(p.W, 0u);
    int bimg0 = 0;
    if constexpr (BUFP) {
        rsW = make_rsrc(p.W, (unsigned)((int64_t)((p.N + 255) / 256 * 256) * p.K * 2));
        if constexpr (AMODE == A_DENSE) {
            rsA = make_rsrc(p.A, (unsigned)((int64_t)p.M * p.lda * 2));
        } else {
            const int ohw = p.cOH * p.cOW, nimg = p.M / ohw;
            const int64_t img = (int64_t)p.cH * p.cW * cld;
            bimg0 = p.bufmode == 2 ? m0 / ohw : 0;
            rsA = make_rsrc(p.A + bimg0 * img, (unsigned)((p.bufmode == 2 ? (nimg - bimg0 < 2 ? nimg - bimg0 : 2) : nimg) * img * 2));
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int m = m0 + 64 * s + 8 * wave + lrow;
        if constexpr (AMODE == A_DENSE) {
            const int mc = m < p.M ? m : p.M - 1;
            a_ptr[s] = p.A + (int64_t)mc * p.lda + cgw * 8;
            a_yx[s] = a_pix0[s] = 0;
            a_voff[s] = (unsigned)((a_ptr[s] - p.A) * 2);
        } else {
            const int ohw = p.cOH * p.cOW;
            const int b = m / ohw, rem = m - b * ohw;
            const int oy = rem / p.cOW, ox = rem - oy * p.cOW;
            a_ptr[s] = p.A + (int64_t)b * p.cH * p.cW * cld + cgw * 8;
            const int iy0 = oy * p.cStride - p.cPad, ix0 = ox * p.cStride - padx;
            a_yx[s] = (int)(m < p.M ? tap_mask(iy0, ix0, p.cH, p.cW) : 0u);
            a_pix0[s] = (iy0 * p.cW + ix0) * cld;
            // byte offset of tap (0, 0) of this lane's pixel inside the resource (the pixel part may be "negative": only in-range taps use the sum)
            a_voff[s] = (unsigned)((int64_t)(b - bimg0) * p.cH * p.cW * cld * 2) + cgw * 16 + (unsigned)(a_pix0[s] * 2);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wave >> 2) * 64 + 32 * j + (wave & 3) * 8 + lrow;
        b_ptr[j] = p.W + (int64_t)(n0 + col_map(r, epi_interleaved<EPI, 2>())) * p.K + cgw * 8;
        b_voff[j] = (unsigned)((b_ptr[j] - p.W) * 2);
    }

    const unsigned *ktab = (const unsigned *)(smem + NBUF * BUF);
    unsigned e_nxt = 0, e_cur = 0;
    // A slots [s_lo, s_hi) of K tile kt_ into buffer `buf` (tiles past the end re-read the last one into a buffer nobody reads any more:
    // the loop stays branch free and every phase uses the same counted wait).  The call that stages slot 0 advances the K-walk cursor.
    auto stage_a = [&](int buf, int kt_, int s_lo, int s_hi) {
        const int ktc = kt_ < nk ? kt_ : nk - 1;
        const int kt = (p.kwrap && ktc >= p.kwrap) ? ktc - p.kwrap : ktc;         // split-fp16 segments re-read A (gemm.h)
        char *base = smem + buf * BUF + wave * 1024;
        unsigned tapoff2 = 0, tapsel = 0;
        if constexpr (AMODE == A_CONV) {
            if (s_lo == 0) {
                e_cur = (unsigned)__builtin_amdgcn_readfirstlane((int)e_nxt);
                e_nxt = ktab[ktc + 1 < nk ? ktc + 1 : nk - 1];
            }
            tapoff2 = ktab_bytes(e_cur); tapsel = ktab_sel(e_cur);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < s_lo || s >= s_hi) continue;
            char *dst = base + s * 8192;
            if constexpr (AMODE == A_DENSE) {
                if constexpr (BUFP) glds16_buf(rsA, (int)a_voff[s], kt * 128, dst);
                else glds16(a_ptr[s] + kt * 64, dst);
            } else {
                const bool ok = ((unsigned)a_yx[s] & tapsel) == tapsel;            // tap_mask: an AND and a compare per DMA
                if constexpr (BUFP) glds16_buf(rsA, (int)(ok ? a_voff[s] + tapoff2 : 0xFFFFFF00u), 0, dst);      // out-of-range offsets read zeros
                else glds16(ok ? a_ptr[s] + (a_pix0[s] + (int)(tapoff2 >> 1)) : p.zero, dst);
            }
        }
    };
    auto stage_b = [&](int buf, int kt_) {
        const int kt = kt_ < nk ? kt_ : nk - 1;
        char *base = smem + buf * BUF + BOFF + ((wave >> 2) * 64 + (wave & 3) * 8) * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (BUFP) glds16_buf(rsW, (int)b_voff[j], kt * 128, base + j * 4096);
            else glds16(b_ptr[j] + kt * 64, base + j * 4096);
        }
    };

    // ---- fragment addressing (gemm8_kernel's): chunk(ks) = (lh ^ fsw) ^ 2 ks  ->  byte offset = c0 ^ (32 ks) ----
    const int li = lane & 31, lh = lane >> 5;
    const int c0 = (lh ^ ((li >> 1) & 7)) * 16;
    const int a_base = (wr * 128 + wm * 64 + li) * 128;     // + rt * 4096
    const int b_base = BOFF + (wn * 64 + li) * 128;         // + j * 4096

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f16x8 fa[2][4], fb0[4], fb1[4];
    const int mxa = p.mx_scale_a * 0x01010101, mxb = p.mx_scale_b * 0x01010101;

    if constexpr (AMODE == A_CONV) {                     // the K walk of the launch (conv_ktab_entry), once per workgroup
        for (int t = tid; t < nk; t += 512) ((unsigned *)(smem + NBUF * BUF))[t] = conv_ktab_entry(p, cld, t);
        __syncthreads();
        e_nxt = ktab[0];
    }
    // prologue: K tiles 0 and 1 (twelve DMAs per thread, in the order the loop certifies them)
    stage_a(0, 0, 0, 3); stage_a(0, 0, 3, 4); stage_b(0, 0);
    stage_a(1, 1, 0, 3); stage_a(1, 1, 3, 4); stage_b(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // K tile 0 landed (this wave's share)
    PB_BAR();
    if (wr == 1) PB_BAR();                                   // stagger the second wave group by one barrier
    if (p.dbg) ts1 = __builtin_readcyclecounter();

    int bcur = 0;                                            // buffer of K tile t
    auto tile = [&](auto fp8_tag, int t) {
        constexpr bool FP8 = decltype(fp8_tag)::value;
        const char *sb = smem + bcur * BUF;
        int bn = bcur + 2;
        bn = bn >= NBUF ? bn - NBUF : bn;                    // buffer of K tile t + 2 (= the one K tile t - 1 was read from)
        // ================= p0 =================
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb0[ks] = *(const f16x8 *)(sb + b_base + (c0 ^ (ks * 32)));
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[rt][ks] = *(const f16x8 *)(sb + a_base + rt * 4096 + (c0 ^ (ks * 32)));
        stage_a(bn, t + 2, 0, 3);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        PB_BAR();
        __builtin_amdgcn_s_setprio(1);
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
                    acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[rt][ks], fb0[ks], acc[rt][0], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        PB_BAR();
        // ================= p1 =================
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fb1[ks] = *(const f16x8 *)(sb + b_base + 4096 + (c0 ^ (ks * 32)));
        stage_a(bn, t + 2, 3, 4);
        stage_b(bn, t + 2);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        PB_BAR();
        __builtin_amdgcn_s_setprio(1);
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
                    acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[rt][ks], fb1[ks], acc[rt][1], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        PB_BAR();
        bcur = bcur + 1 == NBUF ? 0 : bcur + 1;
    };

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

    run_epilogue<EPI, 2, 2, MX>(p, acc, smem, wave, lane, m0 + wr * 128 + wm * 64, n0 + wn * 64, n0);
    if (p.dbg && tid == 0) {                                 // same record as gemm8_kernel's (tools/gemm_stamps.py)
        long long *d = p.dbg + (long long)swz * 8;
        const long long t_issue = __builtin_readcyclecounter();
        d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = t_issue; d[4] = tr0; d[5] = wall_clock64();
        d[6] = t_issue; d[7] = swz;
    }
}

template <int AMODE, int EPI, bool MX, bool BUFP>
int launch_g8n_impl(hipStream_t stream, const GemmArgs &a) {
    constexpr int SMEM = 3 * 49152 + (AMODE == A_CONV ? KTAB_BYTES : 0);
    auto kern = gemm8n_kernel<AMODE, EPI, BUFP, MX>;
    static char name[96];
    if (!name[0]) snprintf(name, sizeof(name), "gemm8n_kernel<%d, %d, %s, %s>", AMODE, EPI, BUFP ? "true" : "false", MX ? "true" : "false");
    pb_gemm_set_last_kernel(name);
    static bool attr_set = false;
    if (!attr_set) {
        PB_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set = true;
    }
    const int tilesM = (a.M + 255) / 256, tilesN = (a.N + 127) / 128;
    hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(512), SMEM, stream, a);
    PB_HIP(hipGetLastError());
    return 0;
}

template <int AMODE, int EPI, bool MX>
int launch_g8n(hipStream_t stream, const GemmArgs &a) {
    GemmArgs b = a;
    b.bufmode = buffer_mode(AMODE, a, 256);
    if (b.bufmode) return launch_g8n_impl<AMODE, EPI, MX, true>(stream, b);
    return launch_g8n_impl<AMODE, EPI, MX, false>(stream, a);
}

}  // namespace
