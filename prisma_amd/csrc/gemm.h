// fp16 MFMA GEMM / implicit-GEMM convolution for gfx950: C[M,N] = A[M,K] * W[N,K]^T (+ fused epilogue).
#pragma once
#include "common.h"

enum { A_DENSE = 0, A_CONV = 1 };
enum { EPI_STD = 0, EPI_RESID = 1, EPI_QKV = 2, EPI_PIXSHUF = 3, EPI_PATCH = 4, EPI_HEAD = 5, EPI_F32 = 6 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SIGMOID = 3, ACT_TANH = 4,
       ACT_GRU_ZR = 5,     // N = 256 = [z | r]: z = sigmoid -> out; r = sigmoid, r * gru_h -> gru_rh (ld 384), nothing to out
       ACT_GRU_Q = 6 };    // N = 128: q = tanh; h = (1 - z) h + z q with z from gru_z (ld 256), h in gru_h (fp32, ld 128); h -> out
enum { TILE_AUTO = 0, TILE_128 = 1, TILE_256 = 2, TILE_N32 = 3, TILE_256x64 = 9, TILE_128x96 = 12 };     // (4, 5, 7, 8, 10: round-1 variants; 11: round 5's 384 x 128 ping-pong
                                                                                       // kernel for N <= 128 - measured slower or equal, removed: EXPERIMENTS.md)

struct GemmArgs {
    // operands: A row-major fp16 [M, lda] (dense) or NHWC image (conv); W fp16 [Npad, K], K % 64 == 0
    const f16 *A = nullptr;
    int64_t lda = 0;
    const f16 *W = nullptr;
    int bufmode = 0;                      // set by launch_gemm: 0 flat LDS-DMA, 1 buffer path over the whole operand, 2 (conv) over a two-image window
    int K = 0, M = 0, N = 0;              // N = columns actually written (multiple of 8)
    // implicit-GEMM convolution: input [B, cH, cW, cC] (cC % 64 == 0), K = KH*KW*cC, rows = (b, oy, ox)
    // kernel KH x cKW (tap = ky * cKW + kx), padding (cPadY, cPadX); cLd = pixel stride in elements (0 = cC), so a
    // channel slice [0, cC) of a wider NHWC buffer can be convolved in place
    int cH = 0, cW = 0, cC = 0, cOH = 0, cOW = 0, cKW = 1, cStride = 1, cPad = 0, cPadX = -1, cLd = 0;
    const f16 *zero = nullptr;            // >= 16 bytes of zeros: source of padded taps / rows >= M
    // conv K order.  0: tap-major - K index = (tap, c): every tap walks the pixel's whole channel extent (cC) before the next tap.
    // 1 (cTapInner, needs cKH): slice-major - K index = (c / 64, tap, c % 64): the KH x KW taps of one 64-half slice run back to back,
    // so the lines a workgroup (and its neighbours on the XCD) touch are re-used by the next 8 tiles while they still sit in the L2;
    // with tap-major order a 384-channel map's per-tap footprint of the tiles in flight on an XCD (~6 MB) exceeds its 4 MB L2 and
    // every tap re-fetches from the fabric.  The weights' K axis is permuted to match (engine_base.hip pack, 128-byte blocks).
    int cTapInner = 0, cKH = 0;
    // split-fp16 operands (precision mode): the K axis is a concatenation of segments [a_hi w_hi | a_lo w_hi | a_hi w_lo]
    // (any subset after the first).  W holds the segments back to back; A holds [hi | lo] (or just hi) and its K index wraps:
    // dense: K tile kt reads A tile (kt >= kwrap ? kt - kwrap : kt); conv: channel cursor c reads channel (c >= kwrap ? c - kwrap : c)
    // of a pixel (cC = channels per tap of the concatenated K axis, cLd = pixel stride of the [hi | lo] image).  0 = off.
    int kwrap = 0;
    // conv: what is added to the channel cursor past kwrap: -kwrap for the wrap above; a map whose fp8 copy sits after the whole
    // pixel's fp16 part ([a16 (Ctot) | a8 (Ctot bytes)], nk16 below) convolved on a channel slice jumps forward instead.
    int kshift = 0;
    int gru_ld = 384;                     // pixel stride (halfs) of gru_rh (the GRU input buffer; 576 with an fp8 copy)
    // MX-fp8 correction segments (split-fp16 mode on shapes that allow it): the operand rows are [fp16 part | fp8 part] in memory - K tiles
    // [0, nk16) hold 64 halfs, tiles [nk16, K / 64) hold 128 OCP e4m3 bytes (same 128 bytes per row and tile, so the staging does
    // not change) and are multiplied with v_mfma_scale_f32_32x32x64_f8f6f4 into the SAME accumulators; the two E8M0 scale bytes undo
    // the power-of-two scalings the fp8 copies were stored with (a_hi 2^pa x w_lo 2^pw, a_lo 2^(pa + 12) x w_hi 2^(pw - 12)).
    // `K` counts 64-half units of the whole row.  nk16 = 0: every tile is fp16.  The pattern repeats every mx_period tiles
    // (0 = the whole K axis): a convolution over split maps [hi fp16 (C) | hi e4m3 (C bytes) | lo e4m3 (C bytes)] per pixel has, per
    // tap, C / 64 fp16 tiles followed by C / 64 fp8 tiles whose bytes [a_hi8 | a_lo8] meet the weights' [w_lo8 | w_hi8].
    int nk16 = 0, mx_period = 0, mx_scale_a = 127, mx_scale_b = 127;
    // with lo_off: the residual parts of split maps are e4m3 - outputs store fp8(hi 2^lo8_pa) at byte lo_off * 2 + n and
    // fp8(lo 2^(lo8_pa + 12)) at byte lo_off * 3 + n of the pixel (lo_off = padded channels), skip tensors are read as hi + lo8
    int lo8 = 0, lo8_pa = 3;
    // EPI_STD direct epilogue: also store fp8(v * o8_scale) - the A operand of a consumer's fp8 segment - at byte offset o8_off of
    // the output row (after its fp16 part); 0 = off
    int o8_off = 0;
    float o8_scale = 16.f;
    // fp16 outputs (out, out2) also store the rounding residual (f16)(v - (float)(f16)v) at element offset +lo_off, and the
    // skip tensors add1 / add2 are read as hi + lo.  0 = off.
    int lo_off = 0;
    // epilogue
    const float *bias = nullptr;          // [N]
    const float *gamma = nullptr;         // LayerScale [N] (EPI_RESID)
    f16 *out = nullptr;                   // act(v) -> out[m*ldo + n]
    f16 *out2 = nullptr;                  // relu(v) copy (same layout)
    const f16 *add1 = nullptr, *add2 = nullptr;   // v += add[m*ldo + n]
    int64_t ldo = 0;
    int act = ACT_NONE;
    float *gru_h = nullptr;        // SepConvGRU fusion (RAFT update.py:55-76), see ACT_GRU_*
    const f16 *gru_z = nullptr;
    f16 *gru_rh = nullptr;
    int pre_relu = 0;                     // EPI_STD: v = relu(acc + bias) before the skip adds
    float *out32 = nullptr;               // EPI_F32: out32[m*ldo + n] = (acc + bias) * scale
    float scale = 1.f;
    float *resid = nullptr;               // fp32 residual stream [rows, ldr]
    int64_t ldr = 0;
    // EPI_QKV: Q,K -> [b, head, ntp, 64]; V -> [b, head, 64, ntp] (transposed); q scaled by qscale
    f16 *q = nullptr, *k = nullptr, *vt = nullptr;
    int ntp = 0, heads = 0, D = 0;
    float qscale = 1.f;
    // EPI_PIXSHUF: rows = (b, y, x) of a [B, ps_h, ps_w] grid; cols = (dy, dx, co); kernel == stride == ps_s
    int ps_s = 1, ps_h = 0, ps_w = 0, ps_co = 0;
    int ps_buf = 0;     // set by launch_gemm: the buffer-addressed pixel-shuffle epilogue applies (gemm_kernels.h pixshuf_epilogue_buf)
    // EPI_PATCH: rows = (b, patch); writes resid[b*ntp + 1 + patch] = v + bias + pos[1 + patch]
    const float *pos = nullptr;
    int ppi = 0;
    // EPI_HEAD: depth[m] = relu(sum_n relu(v + bias)[n] * w2[n] + b2), N == 32
    const float *w2 = nullptr;
    float b2 = 0.f;
    float *depth = nullptr;
    // optional per-block timing stamps (8 x int64 per block): see gemm8_kernel
    long long *dbg = nullptr;
    int stagger = 0;                      // first-wave workgroups sleep ((id >> 3) & 7) * stagger * 64 cycles: see gemm.hip
    // gemm8_kernel (set by its launcher): tiles of the launch - the grid is one persistent workgroup per CU - and whether a tile's
    // first DMAs may be issued in front of the previous tile's epilogue
    int ntiles = 0, prefetch = 1;
    // EPI_RESID on gemm8_kernel: 1 = the accumulators start from the bias alone and X += acc goes out as fire-and-forget fp32 atomic adds
    // (executed in the XCD's L2; every element receives exactly one per launch, so the result is deterministic) instead of a residual
    // tile load in front of the K loop and a store behind it (PB_RESID_ATOMIC)
    int resid_atomic = 0;
    // split-K (the generic 128 x 128 tile only; launch_gemm picks the factor from the tile count when the caller provides sk_ws - the depth engine
    // does for contexts created with max_batch = 1, where a launch has fewer tiles than the chip has CUs): TWO launches of the same kernel.  The
    // first (sk_phase 1, tiles x splitk workgroups) lets slice s accumulate K tiles [s nk / splitk, (s + 1) nk / splitk) and park its fp32
    // accumulators in sk_ws; the second (sk_phase 2, one workgroup per tile, no K tiles) adds the slices up in slice order on top of the bias /
    // residual rows and runs the epilogue.  Deterministic for a given factor; sk_cap = floats sk_ws holds.
    int splitk = 1, sk_phase = 0;
    float *sk_ws = nullptr;
    int64_t sk_cap = 0;
    // packed-channel K axis (conv_walk.h conv_cw3_word; the 128 x 96 tile only): a second copy of the weights whose K axis lists only the cwC real
    // channels of every tap of an mx3 map that carries them as cwPad; launch_gemm switches W / K / nk16 to it when it gives the launch that tile
    // (and sets cwalk = 1: the kernel then builds its chunk table from cwC / cwPad / cwTaps)
    const f16 *Wcw = nullptr;
    int Kcw = 0, nk16cw = 0, cwC = 0, cwPad = 0, cwTaps = 0, cwalk = 0;
    int ablate = 0;                       // PB_GEMM_ABL (timing only, wrong results): 1 no epilogue at all, 2 bare packed-fp16 buffer stores instead of it
};

// Launches the kernel on `stream`.  tile: TILE_AUTO picks from the shape.
int launch_gemm(hipStream_t stream, int amode, int epi, int tile, const GemmArgs &a);

// halo_conv.hip: halo-tiled direct 3x3 convolution 64 -> 64 on mx3 operands (RAFT / GMFlow encoder stage 1).  `a` as EngineBase::conv builds it
// (after set_weights); conv3x3_c64_supported tells whether the shape / layout / epilogue fit (PB_HALO=0 turns the kernel off for A/B runs).
bool conv3x3_c64_supported(const GemmArgs &a);
int launch_conv3x3_c64(hipStream_t stream, const GemmArgs &a);

// Name of the kernel the last launch_gemm call of this thread launched, spelled like the symbol rocprofv3 reports
// ("gemm8_kernel<1, 0, 0, true, false>"): the engines' per-launch timers are keyed by it, so a bench family IS one symbol.
const char *pb_gemm_last_kernel();
// PB_TILE_N96 / pb_set_option "tile_n96" (process-wide): 0 = convolutions with 64 < N <= 96 stay on the 128 x 128 tile; 1 = TILE_AUTO gives them the
// 128 x 96 tile (same bytes); 2 (default) = ... and the packed-channel K axis where the weights carry one (GemmArgs::Wcw: another summation order)
void pb_gemm_set_n96(int mode);
void pb_gemm_set_last_kernel(const char *name);
