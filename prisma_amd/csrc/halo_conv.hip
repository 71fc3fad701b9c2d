// Halo-tiled direct 3x3 convolution, 64 -> 64 channels, stride 1, over split maps with e4m3 residual parts (gemm.h lo8) - the
// half-resolution stage of RAFT's encoders (bands/raft/extractor.py:118-192 layer1: four 3x3 convolutions per encoder on maps of
// (H / 2) x (W / 2) x 64; the same layers of GMFlow's CNNEncoder, bands/gmflow/backbone.py:66-72).
//
// Why not the implicit GEMM (gemm_kernels.h): with N = 64 a 256-pixel tile does 2.1 MFLOP per K tile against 40 KB of LDS-DMA, and every
// input pixel is staged nine times (once per tap).  Measured (profiles/r02g): 2.46 ms per launch on 9.4 M pixels = 282 TF/s, 2.4x the
// algorithmic HBM bytes - the kernel is bound by the L2 -> LDS stream, not by the matrix pipe or by HBM.  Here a workgroup stages an
// 8 x 32 output tile's 10 x 34 input pixels ONCE (1.33x the tile's own pixels instead of 9x) and walks the nine taps out of the LDS:
//   * LDS: input tile 340 pixels x 256 B ([hi fp16 (64) | hi e4m3 (64) | lo e4m3 (64)] per pixel, the 16-byte chunk index XOR-swizzled
//     with (pixel & 15): the fragment reads of 32 consecutive pixels are bank-conflict free) + two 16 KB weight slabs (one tap each:
//     64 output channels x [w_hi fp16 | w_lo e4m3 | w_hi e4m3]), double buffered - 117 KB, one workgroup per CU;
//   * everything arrives by LDS-DMA through buffer resources (`buffer_load ... lds`, out-of-range offsets return zeros: the zero padding
//     of the image border and of ragged tiles costs nothing); the swizzle is applied on the SOURCE address, per lane and loop invariant;
//   * 4 waves x (2 row segments of 32 pixels) x 64 channels: per tap 16 fp16 MFMAs (a_hi w_hi) and 8 MX-scaled fp8 MFMAs
//     (a_hi8 w_lo8 + a_lo8 w_hi8) into the same fp32 accumulators - the arithmetic of the implicit-GEMM path (gemm.h nk16 / mx_period);
//   * persistent over tiles: the next tile's input and first weight slab are requested before the epilogue of the current one, so the
//     HBM latency of a tile hides under the previous tile's stores;
//   * epilogue as EPI_STD's: bias, ReLU before / after, one skip tensor (read as hi + lo8), outputs as [hi | hi8 | lo8].  Weight rows are
//     fed interleaved (LDS row tn * 32 + j <- output channel 2 j + tn) so a lane owns two adjacent channels and half a wave stores one
//     contiguous 128-byte line per pixel (gemm_kernels.h "interleaved output columns").
#include "gemm.h"

#include <stdlib.h>

#include <vector>

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 8, TW = 32, IW = TW + 2, IH = TH + 2, NPIX = IW * IH;      // 340 input pixels per tile
constexpr int PXB = 256;                                                       // bytes per pixel / per (channel row, tap) of the weights
constexpr int IN_BYTES = NPIX * PXB, W_BYTES = 64 * PXB, SMEM = IN_BYTES + 3 * W_BYTES;
constexpr int IN_DMAS = NPIX / 4;                                              // 85 DMA instructions of 1 KB (4 pixels) per tile
static_assert(NPIX % 4 == 0, "a DMA instruction carries 4 pixels");

__device__ __forceinline__ i32x8 cat2(f16x8 a, f16x8 b) {
    union { f16x8 h[2]; i32x8 v; } u;
    u.h[0] = a; u.h[1] = b;
    return u.v;
}

// PIPE = 0: one weight slab ahead, a tap's fragments read right before its MFMAs (the first version: 2.11 ms per launch against the implicit
// GEMM's 2.48).  PIPE = 1: two slabs ahead on a three-slab ring, and tap t + 1's fragments are read from the LDS under tap t's MFMAs (with
// one wave per SIMD nothing else hides the ~500 cycles the 128 KB of fragment reads per tap take); the next tile's input is requested
// before the LAST tap's MFMAs.
template <int PIPE>
__global__ __launch_bounds__(256) void conv3x3_c64_mx_kernel(const GemmArgs p, int tilesX, int tilesPerImg, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *s_in = smem, *s_w = smem + IN_BYTES;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int H = p.cH, W = p.cW;
    const int64_t img_bytes = (int64_t)H * W * PXB;

    // ---- loop-invariant DMA geometry ----
    // input: instruction j of this wave carries pixels q = 4 (wave + 4 j) + (lane >> 4) = q0 + 16 j; q & 15 (the swizzle key) does not change
    const int q0 = 4 * wave + (lane >> 4);
    const int in_chunk = (lane & 15) ^ (q0 & 15);                       // logical 16-byte chunk this lane fetches
    int in_row0 = q0 / IW, in_col0 = q0 - in_row0 * IW;
    // weights: instruction j carries LDS rows r = q0 + 16 j (same lane pattern); LDS row r holds output channel 2 (r & 31) + (r >> 5)
    unsigned w_voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = q0 + 16 * j, n = 2 * (r & 31) + (r >> 5);
        w_voff[j] = (unsigned)(n * p.K * 2 + (((lane & 15) ^ (r & 15)) * 16));
    }
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.W, (unsigned)(64 * p.K * 2));
    auto stage_weights = [&](int tap, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_buf(rsW, (int)w_voff[j], tap * PXB, s_w + buf * W_BYTES + (wave + 4 * j) * 1024);
    };
    auto stage_input = [&](int tile) {
        const int b = tile / tilesPerImg, t = tile - b * tilesPerImg;
        const int ty = t / tilesX, tx = t - ty * tilesX;
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc((const char *)p.A + (int64_t)b * img_bytes, (unsigned)img_bytes);
        int row = in_row0, col = in_col0;
        for (int j = 0; 4 * j + wave < IN_DMAS; ++j) {
            const int iy = y0 + row, ix = x0 + col;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const unsigned off = ok ? (unsigned)((iy * W + ix) * PXB + in_chunk * 16) : 0xFFFFFF00u;      // out of range: the load returns zeros
            glds16_buf(rs, (int)off, 0, s_in + (wave + 4 * j) * 1024);
            col += 16;
            if (col >= IW) { col -= IW; ++row; }
        }
    };

    // ---- fragment addressing ----
    // A: output row segment s = 2 wave + tm (tile row), pixel li; tap (ky, kx) reads input pixel q = (s + ky) * 34 + li + kx
    // B: LDS row tn * 32 + li
    int b_off[2], b_key[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) { const int r = tn * 32 + li; b_off[tn] = r * PXB; b_key[tn] = r & 15; }
    const int sa = p.mx_scale_a * 0x01010101, sb = p.mx_scale_b * 0x01010101;

    // per-lane epilogue constants: channels n = 2 li, 2 li + 1
    const int n = 2 * li;
    const float b0 = p.bias ? p.bias[n] : 0.f, b1 = p.bias ? p.bias[n + 1] : 0.f;
    const float shi = __builtin_ldexpf(1.f, p.lo8_pa), slo = __builtin_ldexpf(1.f, p.lo8_pa + 12), inv_lo = __builtin_ldexpf(1.f, -(p.lo8_pa + 12));

    // fragments of one tap: A = input pixels of this wave's two row segments shifted by (ky, kx), B = the tap's weight slab
    auto load_frags = [&](int tap, const char *sw, f16x8 (&af)[2][8], f16x8 (&bf)[2][8]) {
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int q = (2 * wave + tm + ky) * IW + li + kx;
            const char *base = s_in + q * PXB;
            const int key = q & 15;
#pragma unroll
            for (int c = 0; c < 8; ++c) af[tm][c] = *(const f16x8 *)(base + (((2 * c + lh) ^ key) * 16));
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int c = 0; c < 8; ++c) bf[tn][c] = *(const f16x8 *)(sw + b_off[tn] + (((2 * c + lh) ^ b_key[tn]) * 16));
    };
    // chunks 2 c + lh, c = 0..3: the four fp16 k-steps (a_hi w_hi); c = 4..7: the 128 e4m3 bytes [hi8 | lo8] x [w_lo8 | w_hi8] as two
    // MX-scaled MFMAs of 64 bytes each (fragments c, c + 1 form the 32-byte operand; any byte order shared by A and B is valid)
    auto mfmas = [&](f32x16 (&acc)[2][2], const f16x8 (&af)[2][8], const f16x8 (&bf)[2][8]) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tm][c], bf[tn][c], acc[tm][tn], 0, 0, 0);
#pragma unroll
        for (int c = 4; c < 8; c += 2)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat2(af[tm][c], af[tm][c + 1]), cat2(bf[tn][c], bf[tn][c + 1]),
                                                                                 acc[tm][tn], 0, 0, 0, sa, 0, sb);
    };

    int tile = blockIdx.x;
    if (tile < ntiles) {
        stage_input(tile);
        stage_weights(0, 0);
        if (PIPE) stage_weights(1, 1);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int next = tile + gridDim.x;
        if constexpr (PIPE == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's share of the input tile and of tap 0's weights (and its old stores)
            __syncthreads();
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) stage_weights(tap + 1, (tap + 1) & 1);
                f16x8 af[2][8], bf[2][8];
                load_frags(tap, s_w + (tap & 1) * W_BYTES, af, bf);
                mfmas(acc, af, bf);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // next tap's weights (this wave's pieces)
                __syncthreads();                                      // ... everyone's; and everyone is done with this tap's slab
            }
            // the input tile and both weight slabs are dead: request the next tile before storing this one
            if (next < ntiles) { stage_input(next); stage_weights(0, 0); }
        } else {
            f16x8 af[2][2][8], bf[2][2][8];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // slab of tap t: t % 3.  On entry the DMAs in flight are those of tap + 1's slab (tap 0: also the input tile and slab 0)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                 // slab tap + 1 complete; every wave has read tap's fragments, so slab (tap + 2) % 3 (= tap - 1's) is free
                if (tap == 0) load_frags(0, s_w, af[0], bf[0]);
                if (tap + 2 < 9) stage_weights(tap + 2, (tap + 2) % 3);
                if (tap == 8 && next < ntiles) {         // the input tile was last read for tap 8's fragments (under tap 7): request the next tile now
                    stage_input(next);
                    stage_weights(0, 0);
                    stage_weights(1, 1);
                }
                if (tap + 1 < 9) load_frags(tap + 1, s_w + ((tap + 1) % 3) * W_BYTES, af[(tap + 1) & 1], bf[(tap + 1) & 1]);
                mfmas(acc, af[tap & 1], bf[tap & 1]);
            }
        }

        // ---- epilogue (EPI_STD semantics, gemm_kernels.h direct_epilogue_f16_impl LOM = 2) ----
        const int b = tile / tilesPerImg, t = tile - b * tilesPerImg;
        const int ty = t / tilesX, tx = t - ty * tilesX;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int y = ty * TH + 2 * wave + tm;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float v0[8], v1[8];
                int64_t off[8];
                bool ok[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = hf * 8 + q;
                    const int x = tx * TW + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    ok[q] = y < H && x < W;
                    const int xc = x < W ? x : W - 1, yc = y < H ? y : H - 1;
                    off[q] = (((int64_t)b * H + yc) * W + xc) * p.ldo;
                    v0[q] = acc[tm][0][r] + b0; v1[q] = acc[tm][1][r] + b1;
                }
                if (p.pre_relu) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
                }
                if (p.add1) {
                    f16x2 a[8];
                    unsigned short l8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        a[q] = *(const f16x2 *)(p.add1 + off[q] + n);
                        l8[q] = *(const unsigned short *)((const char *)(p.add1 + off[q]) + 3 * p.lo_off + n);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const f32x2 l = __builtin_amdgcn_cvt_pk_f32_fp8((int)l8[q], false);
                        v0[q] += (float)a[q][0] + l[0] * inv_lo; v1[q] += (float)a[q][1] + l[1] * inv_lo;
                    }
                }
                if (p.act == ACT_RELU) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (!ok[q]) continue;
                    const f16 h0 = (f16)v0[q], h1 = (f16)v1[q];
                    f16x2 o; o[0] = h0; o[1] = h1;
                    *(f16x2 *)(p.out + off[q] + n) = o;
                    char *pb = (char *)(p.out + off[q]);
                    *(unsigned short *)(pb + 2 * p.lo_off + n) = pb_fp8x2((float)h0 * shi, (float)h1 * shi);
                    *(unsigned short *)(pb + 3 * p.lo_off + n) = pb_fp8x2((v0[q] - (float)h0) * slo, (v1[q] - (float)h1) * slo);
                }
            }
        }
    }
}

// ---- second geometry: 8 x 16 output tiles, two workgroups per CU --------------------------------------------------------------
// The 8 x 32 kernel above holds 117-136 KB of LDS: one workgroup per CU, so nothing runs while a tile's 87 KB arrive or its 64 KB leave -
// and the layer is HBM-bound (5.9 GB per launch), i.e. those phases ARE the kernel.  Here a workgroup takes 8 x 16 outputs (10 x 18 input
// pixels: 45 KB, halo 1.41x; two weight slabs), 77 KB in all, and TWO workgroups share a CU: one's loads and stores run under the other's
// MFMAs.  A wave owns 32 pixels = two rows of 16 (MFMA row li -> row li >> 4, column li & 15); the swizzle key is the pixel's COLUMN in
// the input tile (mod 16), so the two rows a fragment read touches use the same keys at different columns and stay conflict free.
constexpr int TH2 = 8, TW2 = 16, IW2 = TW2 + 2, IH2 = TH2 + 2, NPIX2 = IW2 * IH2;       // 180 input pixels
constexpr int IN2_BYTES = NPIX2 * PXB, SMEM2 = IN2_BYTES + 2 * W_BYTES, IN2_DMAS = NPIX2 / 4;
static_assert(NPIX2 % 4 == 0, "a DMA instruction carries 4 pixels");

__global__ __launch_bounds__(256, 2) void conv3x3_c64_mx2_kernel(const GemmArgs p, int tilesX, int tilesPerImg, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *s_in = smem, *s_w = smem + IN2_BYTES;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int H = p.cH, W = p.cW;
    const int64_t img_bytes = (int64_t)H * W * PXB;

    const int q0 = 4 * wave + (lane >> 4);                          // instruction j of this wave carries pixel q0 + 16 j
    const int in_row0 = q0 / IW2, in_col0 = q0 - in_row0 * IW2;
    unsigned w_voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = q0 + 16 * j, n = 2 * (r & 31) + (r >> 5);
        w_voff[j] = (unsigned)(n * p.K * 2 + (((lane & 15) ^ (r & 15)) * 16));
    }
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.W, (unsigned)(64 * p.K * 2));
    auto stage_weights = [&](int tap, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_buf(rsW, (int)w_voff[j], tap * PXB, s_w + buf * W_BYTES + (wave + 4 * j) * 1024);
    };
    auto stage_input = [&](int tile) {
        const int b = tile / tilesPerImg, t = tile - b * tilesPerImg;
        const int ty = t / tilesX, tx = t - ty * tilesX;
        const int y0 = ty * TH2 - 1, x0 = tx * TW2 - 1;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc((const char *)p.A + (int64_t)b * img_bytes, (unsigned)img_bytes);
        int row = in_row0, col = in_col0;
        for (int j = 0; 4 * j + wave < IN2_DMAS; ++j) {
            const int iy = y0 + row, ix = x0 + col;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int chunk = (lane & 15) ^ (col & 15);              // swizzle key = the pixel's column in the tile
            const unsigned off = ok ? (unsigned)((iy * W + ix) * PXB + chunk * 16) : 0xFFFFFF00u;
            glds16_buf(rs, (int)off, 0, s_in + (wave + 4 * j) * 1024);
            col += 16;
            if (col >= IW2) { col -= IW2; ++row; }
        }
    };

    int b_off[2], b_key[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) { const int r = tn * 32 + li; b_off[tn] = r * PXB; b_key[tn] = r & 15; }
    const int sa = p.mx_scale_a * 0x01010101, sb = p.mx_scale_b * 0x01010101;
    const int arow = 2 * wave + (li >> 4), acol = li & 15;          // this lane's A row: tile row, tile column
    const int n = 2 * li;
    const float b0 = p.bias ? p.bias[n] : 0.f, b1 = p.bias ? p.bias[n + 1] : 0.f;
    const float shi = __builtin_ldexpf(1.f, p.lo8_pa), slo = __builtin_ldexpf(1.f, p.lo8_pa + 12), inv_lo = __builtin_ldexpf(1.f, -(p.lo8_pa + 12));

    // optional cycle stamps (PB_HALO_DBG=1, launcher): per workgroup and tile 0..3: [loop top, input landed, taps done, epilogue issued]
    long long *dbg = p.dbg ? p.dbg + (long long)blockIdx.x * 16 : nullptr;
    int tcount = 0;
    int tile = blockIdx.x;
    if (tile < ntiles) { stage_input(tile); stage_weights(0, 0); }
    // Pipeline of a tile (round 3, second form): the tap loop is unrolled (its address arithmetic folds: 15.2 -> 14.4 ms per step); tap t + 1's A
    // fragments are read under tap t's MFMAs, so after tap 7 the input tile is dead and the NEXT tile's input and tap-0 weights are requested at
    // the start of tap 8 instead of after it; the weight slab of a tile's tap 0 alternates (wpar) so that request has a free slab, and the
    // epilogue assembles its records in the slab tap 8 just finished with.
    int wpar = 0;                                               // slab of this tile's tap 0
    for (; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        if (dbg && tid == 0 && tcount < 4) dbg[tcount * 4 + 0] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (dbg && tid == 0 && tcount < 4) dbg[tcount * 4 + 1] = __builtin_readcyclecounter();
        const int next = tile + gridDim.x;
        f16x8 af[2][8];
        auto load_a = [&](int tap, f16x8 (&dst)[8]) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int q = (arow + ky) * IW2 + acol + kx, key = (acol + kx) & 15;
            const char *base = s_in + q * PXB;
#pragma unroll
            for (int c = 0; c < 8; ++c) dst[c] = *(const f16x8 *)(base + (((2 * c + lh) ^ key) * 16));
        };
        load_a(0, af[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int slot = (wpar + tap) & 1;
            if (tap + 1 < 9) stage_weights(tap + 1, slot ^ 1);
            else if (next < ntiles) { stage_input(next); stage_weights(0, slot ^ 1); }     // every wave holds tap 8's A fragments: the input tile is dead
            const char *sw = s_w + slot * W_BYTES;
            f16x8 bf[2][8];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int c = 0; c < 8; ++c) bf[tn][c] = *(const f16x8 *)(sw + b_off[tn] + (((2 * c + lh) ^ b_key[tn]) * 16));
            if (tap + 1 < 9) load_a(tap + 1, af[(tap + 1) & 1]);
            f16x8 (&a)[8] = af[tap & 1];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], bf[tn][c], acc[tn], 0, 0, 0);
#pragma unroll
            for (int c = 4; c < 8; c += 2)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat2(a[c], a[c + 1]), cat2(bf[tn][c], bf[tn][c + 1]), acc[tn], 0, 0, 0, sa, 0, sb);
            if (tap + 1 < 9) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // next tap's weights (this wave's pieces)
                __syncthreads();                                       // ... everyone's; and everyone has read this tap's slab and the next tap's A fragments
            } else {
                // after tap 8 only the LDS reads matter (the slab becomes the epilogue's staging area); the next tile's DMAs stay in flight - a bare
                // barrier, since __syncthreads() would drain them
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (dbg && tid == 0 && tcount < 4) dbg[tcount * 4 + 2] = __builtin_readcyclecounter();
        const int stg_slot = wpar;                                  // tap 8's slab ((wpar + 8) & 1)
        wpar ^= 1;                                                  // the next tile's tap 0 went to the other one

        const int b = tile / tilesPerImg, t = tile - b * tilesPerImg;
        const int ty = t / tilesX, tx = t - ty * tilesX;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            float v0[8], v1[8];
            int64_t off[8];
            bool ok[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = hf * 8 + q;
                const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;       // accumulator row = pixel of this wave's 2 x 16 block
                const int y = ty * TH2 + 2 * wave + (i >> 4), x = tx * TW2 + (i & 15);
                ok[q] = y < H && x < W;
                const int xc = x < W ? x : W - 1, yc = y < H ? y : H - 1;
                off[q] = (((int64_t)b * H + yc) * W + xc) * p.ldo;
                v0[q] = acc[0][r] + b0; v1[q] = acc[1][r] + b1;
            }
            if (p.pre_relu) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
            }
            if (p.add1) {
                f16x2 a[8];
                unsigned short l8[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    a[q] = *(const f16x2 *)(p.add1 + off[q] + n);
                    l8[q] = *(const unsigned short *)((const char *)(p.add1 + off[q]) + 3 * p.lo_off + n);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x2 l = __builtin_amdgcn_cvt_pk_f32_fp8((int)l8[q], false);
                    v0[q] += (float)a[q][0] + l[0] * inv_lo; v1[q] += (float)a[q][1] + l[1] * inv_lo;
                }
            }
            if (p.act == ACT_RELU) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
            }
            // Stores: this half is ONE tile row - 16 consecutive pixels, i.e. 4 KB of contiguous [hi | hi8 | lo8] records.  Written from the
            // accumulator layout they would be 24 narrow store instructions per lane (4-byte hi pairs, 2-byte e4m3 pairs): measured, the epilogue
            // took 10.4k of a tile's 24.5k cycles.  The records are assembled in the LDS (this wave's 4 KB of the idle weight slab 1) and leave as
            // four 16-byte stores per lane, 1 KB contiguous per instruction.  (The slab is the one tap 8 read: the other one is receiving the next tile's tap 0.)
            char *stg = s_w + stg_slot * W_BYTES + wave * 4096;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = hf * 8 + q;
                const int px = ((r & 3) + 8 * (r >> 2) + 4 * lh) & 15;
                const f16 h0 = (f16)v0[q], h1 = (f16)v1[q];
                f16x2 o; o[0] = h0; o[1] = h1;
                char *rec = stg + px * PXB;
                *(f16x2 *)(rec + 2 * n) = o;
                *(unsigned short *)(rec + 128 + n) = pb_fp8x2((float)h0 * shi, (float)h1 * shi);
                *(unsigned short *)(rec + 192 + n) = pb_fp8x2((v0[q] - (float)h0) * slo, (v1[q] - (float)h1) * slo);
            }
            {
                const int y = ty * TH2 + 2 * wave + hf, x0 = tx * TW2;
                char *row = (char *)(p.out + (((int64_t)b * H + (y < H ? y : H - 1)) * W + x0) * p.ldo);
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const int c = lane + 64 * c4;                    // 16-byte chunk of the row: pixel c >> 4
                    const f16x8 v = *(const f16x8 *)(stg + c * 16);
                    if (y < H && x0 + (c >> 4) < W) *(f16x8 *)(row + c * 16) = v;
                }
            }
        }
        if (dbg && tid == 0 && tcount < 4) dbg[tcount * 4 + 3] = __builtin_readcyclecounter();
        ++tcount;
    }
}

}  // namespace

// Whether launch_conv3x3_c64 can run this convolution (after EngineBase::set_weights): 3x3 / stride 1 / pad 1, 64 -> 64 channels on mx3
// operands (maps [hi | hi8 | lo8] with 128-half pixels, tap-major weights [w_hi | w_lo8 | w_hi8]), plain EPI_STD epilogue
bool conv3x3_c64_supported(const GemmArgs &a) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("PB_HALO"); on = e ? atoi(e) : 1; }
    return on && a.cKW == 3 && a.cStride == 1 && a.cPad == 1 && (a.cPadX < 0 || a.cPadX == 1) && a.N == 64 && a.cLd == 128 && a.ldo == 128 && a.lo_off == 64 && a.lo8 &&
           a.nk16 == 1 && a.mx_period == 2 && a.cC == 128 && a.K == 9 * 128 && !a.cTapInner && !a.kwrap && !a.out2 && !a.add2 && !a.gru_h && !a.o8_off &&
           (a.act == ACT_NONE || a.act == ACT_RELU) && a.cOH == a.cH && a.cOW == a.cW && (int64_t)a.cH * a.cW * PXB < (1LL << 32) - (1 << 20);
}

int launch_conv3x3_c64(hipStream_t stream, const GemmArgs &a) {
    PB_CHECK(conv3x3_c64_supported(a), -1, "conv3x3_c64: unsupported shape / layout");
    static bool attr_set = false;
    if (!attr_set) {
        PB_HIP(hipFuncSetAttribute((const void *)conv3x3_c64_mx_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        PB_HIP(hipFuncSetAttribute((const void *)conv3x3_c64_mx_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set = true;
    }
    static int mode = -1;
    if (mode < 0) { const char *e = getenv("PB_HALO"); mode = e ? atoi(e) : 3; }      // 1: 8 x 32 unpipelined, 2: 8 x 32 pipelined, 3 (default): 8 x 16, two workgroups per CU
    const bool small = mode >= 3;
    const int tilesX = small ? (a.cW + TW2 - 1) / TW2 : (a.cW + TW - 1) / TW, tilesY = (a.cH + TH - 1) / TH, nimg = a.M / (a.cH * a.cW);
    const int ntiles = tilesX * tilesY * nimg;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        PB_HIP(hipGetDevice(&dev));
        PB_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (small) {
        static bool attr2 = false;
        if (!attr2) {
            PB_HIP(hipFuncSetAttribute((const void *)conv3x3_c64_mx2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2));
            attr2 = true;
        }
        const int grid2 = ntiles < 2 * ncu ? ntiles : 2 * ncu;
        pb_gemm_set_last_kernel("conv3x3_c64_mx2_kernel");
        static int dbg_left = -1;
        if (dbg_left < 0) { const char *e = getenv("PB_HALO_DBG"); dbg_left = e ? atoi(e) : 0; }
        if (dbg_left > 0 && grid2 == 2 * ncu) {          // phase stamps of one launch (experiment switch; synchronises)
            --dbg_left;
            long long *d = nullptr;
            PB_HIP(hipMalloc((void **)&d, (size_t)grid2 * 16 * 8));
            PB_HIP(hipMemsetAsync(d, 0, (size_t)grid2 * 16 * 8, stream));
            GemmArgs b = a;
            b.dbg = d;
            hipLaunchKernelGGL(conv3x3_c64_mx2_kernel, dim3(grid2), dim3(256), SMEM2, stream, b, tilesX, tilesX * tilesY, ntiles);
            PB_HIP(hipStreamSynchronize(stream));
            std::vector<long long> h((size_t)grid2 * 16);
            PB_HIP(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
            PB_HIP(hipFree(d));
            double w = 0, tp = 0, ep = 0, tot = 0;
            int cnt = 0;
            for (int g = 0; g < grid2; ++g)
                for (int t = 1; t < 3; ++t) {            // tiles 1 and 2 of every workgroup (steady state)
                    const long long *x = h.data() + (size_t)g * 16 + t * 4;
                    if (!x[3] || !x[4]) continue;
                    w += x[1] - x[0]; tp += x[2] - x[1]; ep += x[3] - x[2]; tot += x[4] - x[0]; ++cnt;
                }
            if (cnt) fprintf(stderr, "[halo dbg] per tile (cycles, mean of %d): wait for input %.0f, nine taps %.0f, request + epilogue %.0f, tile period %.0f\n", cnt, w / cnt, tp / cnt, ep / cnt, tot / cnt);
            return 0;
        }
        hipLaunchKernelGGL(conv3x3_c64_mx2_kernel, dim3(grid2), dim3(256), SMEM2, stream, a, tilesX, tilesX * tilesY, ntiles);
        PB_HIP(hipGetLastError());
        return 0;
    }
    const int grid = ntiles < ncu ? ntiles : ncu;
    const int pipe = mode != 1;
    pb_gemm_set_last_kernel(pipe ? "conv3x3_c64_mx_kernel<1>" : "conv3x3_c64_mx_kernel<0>");
    if (pipe) hipLaunchKernelGGL(conv3x3_c64_mx_kernel<1>, dim3(grid), dim3(256), SMEM, stream, a, tilesX, tilesX * tilesY, ntiles);
    else hipLaunchKernelGGL(conv3x3_c64_mx_kernel<0>, dim3(grid), dim3(256), SMEM, stream, a, tilesX, tilesX * tilesY, ntiles);
    PB_HIP(hipGetLastError());
    return 0;
}
