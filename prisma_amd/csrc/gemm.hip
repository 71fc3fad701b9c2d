// fp16 MFMA GEMM / implicit-GEMM convolution: argument checks, tile choice and dispatch to the kernel instantiations
// (gemm_kernels.h holds the kernels; gemm_i*.hip instantiate them, one group per translation unit).
#include "gemm.h"

#include <stdlib.h>

#include <algorithm>

int pb_gemm_dense_std_f16(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_dense_std_mx(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_dense_resid_f16(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_dense_resid_mx(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_dense_qkv_f16(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_dense_qkv_mx(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_dense_pixshuf_f16(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_dense_pixshuf_mx(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_dense_patch_f16(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_dense_f32_f16(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_conv_f32_f16(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_conv_std_f16(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_conv_std_mx(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_conv_pixshuf_f16(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_conv_pixshuf_mx(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_conv_head_f16(hipStream_t s, int tile, const GemmArgs &a);
int pb_gemm_conv_head_mx(hipStream_t s, int tile, const GemmArgs &a);

static int g_n96 = -1;                          // -1: not read yet (PB_TILE_N96, default 2: gemm.h pb_gemm_set_n96)
void pb_gemm_set_n96(int mode) { g_n96 = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
static thread_local const char *g_last_kernel = "";
const char *pb_gemm_last_kernel() { return g_last_kernel; }
void pb_gemm_set_last_kernel(const char *name) { g_last_kernel = name; }

int launch_gemm(hipStream_t stream, int amode, int epi, int tile, const GemmArgs &a_in) {
    GemmArgs a = a_in;
    PB_CHECK(a.K > 0 && a.K % 64 == 0, -1, "gemm: K=%d must be a positive multiple of 64", a.K);
    PB_CHECK(a.M > 0 && a.N > 0 && a.N % 8 == 0, -1, "gemm: bad M=%d N=%d", a.M, a.N);
    if (amode == A_CONV) PB_CHECK(a.cC % 64 == 0 && a.cLd % 8 == 0 && a.zero, -1, "conv: channels %d (x64) / pixel stride %d (x8)", a.cC, a.cLd);
    // gemm_kernel tests a tap against an 8 + 8 bit mask of the pixel (gemm_kernels.h tap_mask)
    if (amode == A_CONV) PB_CHECK(a.K / 64 <= 512 && a.kshift % 8 == 0 && (int64_t)8 * a.cW * (a.cLd ? a.cLd : a.cC) * 2 < (1LL << 30), -1,
                                  "conv: K walk table (K = %d, kshift = %d, row of %d pixels)", a.K, a.kshift, a.cW);
    if (amode == A_CONV) PB_CHECK(a.cKW >= 1 && a.cKW <= 8 && a.K / (a.cKW * a.cC) <= 8, -1, "conv: kernel %d x %d (up to 8 x 8)", a.K / (a.cKW * a.cC), a.cKW);
    {   // experiment switch: PB_GEMM_STAGGER=<percent> de-phases the CUs by that share of 1/8 tile period per step
        static int env_st = -2;
        if (env_st == -2) { const char *e = getenv("PB_GEMM_STAGGER"); env_st = e ? atoi(e) : -1; }
        const int nk = a.K / 64;
        const int est = (3200 * nk + 24000) / 8 / 64;
        a.stagger = env_st > 0 ? est * env_st / 100 : 0;    // measured: no gain (0.97-1.0x), so off unless requested
    }
    if (epi == EPI_PIXSHUF) {
        // buffer-addressed pixel-shuffle epilogue (gemm_kernels.h pixshuf_epilogue_buf): at most TM <= 4 grid rows per wave tile, and the
        // window a wave addresses (128 rows x s pixels + 4 wraps of s (s - 1) ps_w pixels) well inside the 2 GB a resource spans
        static int env_pb = -1;
        if (env_pb < 0) env_pb = pb_env_int("PB_PIXSHUF_BUF", 1);
        const int64_t pbytes = a.ldo * 2, window = (int64_t)(128 * a.ps_s + 4 * a.ps_s * (a.ps_s - 1) * (int64_t)a.ps_w) * pbytes;
        a.ps_buf = env_pb && a.ps_w >= 32 && a.ps_co % 64 == 0 && window < (1LL << 30) ? 1 : 0;
    }
    if (tile == TILE_AUTO) {
        // 256x256 needs wide N and enough tiles to fill 256 CUs; the q/k/v split needs D % BN == 0
        // PB_WIDE_MIN=<tiles>: the fewest 256 x 256 tiles a launch must have to take the ping-pong kernel; default 256 = one per CU.  A one-frame
        // context (the caller lent a split-K workspace: gemm.h sk_ws) takes it from 128 tiles on: fc1 of a 720p frame (144 tiles) and the head's
        // full-resolution convolutions run one round of 256 x 256 tiles faster than 1.25 rounds of 128 x 128 ones (7.11 -> 6.93 ms per frame,
        // profiles/r06p_latency_knobs.txt; 100 - qkv as well - is slower again).  Same bits either way: every tile shape walks K in one order.
        static int wide_tiles = -1;
        if (wide_tiles < 0) wide_tiles = pb_env_int("PB_WIDE_MIN", 256);
        const int wide_need = a.sk_ws ? std::min(wide_tiles, 128) : wide_tiles;
        const bool wide = a.N % 256 == 0 && (int64_t)(a.M / 256) * (a.N / 256) >= wide_need;
        tile = wide && (epi != EPI_QKV || a.D % 256 == 0) ? TILE_256 : TILE_128;
        // N = 192 (RAFT's convc2): two 128-wide tiles of which the second is half empty, or one 256-wide tile that is a quarter empty - the
        // ping-pong kernel's K loop runs ~2460 cycles per 256 x 256 x 64 against the generic tile's ~1000 per 128 x 128 x 64.  Measured (r04p, one box):
        // flow band 147.4 -> 145.3 ms with N = 192 on the wide tile, 157.8 with N = 128 on it too.  PB_TILE_WIDE=<min columns of the last tile>, 0 = off
        static int wide_min = -1;
        if (wide_min < 0) { const char *e = getenv("PB_TILE_WIDE"); wide_min = e ? atoi(e) : 192; }
        if (tile == TILE_128 && epi == EPI_STD && wide_min > 0 && a.N % 256 >= wide_min && (int64_t)(a.M / 256) * ((a.N + 255) / 256) >= 256) tile = TILE_256;
        // N <= 64 (RAFT / ResNet stems and first stages): a 128-wide tile spends half its MFMAs and B loads on padding columns
        static int n64_tile = -1;
        if (n64_tile < 0) { const char *e = getenv("PB_TILE_N64"); n64_tile = e ? atoi(e) : TILE_256x64; }
        if (tile == TILE_128 && epi == EPI_STD && a.N <= 64 && n64_tile != TILE_128) tile = n64_tile;
        // (round 5 built a 384 x 128 ping-pong kernel for 64 < N <= 128 - K loop at 70 % of the matrix pipe against this tile's ~45 % - whose one
        // workgroup per CU exposed its set-up and epilogue: 35.5 against 34.6 ms on the bands' launches; under two-band overlap, round 6, +0.2 ... 0.8 %
        // of a step (profiles/r06b_overlap_n128.txt, r06c_bench_n128_line.json).  Not a gain worth 1300 lines and two translation units: removed.)
        // 64 < N <= 96 (RAFT / GMFlow encoder stage 2 carries 96 channels): the 128 x 96 tile - a quarter fewer MFMAs, B rows and fragment reads than
        // the 128-wide tile spends on 32 padding columns; its third column block has a plain single-column epilogue (gemm_kernels.h single_col_epilogue)
        if (g_n96 < 0) pb_gemm_set_n96(pb_env_int("PB_TILE_N96", 2));
        if (tile == TILE_128 && g_n96 && amode == A_CONV && epi == EPI_STD && a.N > 64 && a.N <= 96 && (a.act == ACT_NONE || a.act == ACT_RELU) && !a.out2 && !a.o8_off)
            tile = TILE_128x96;
        static int small_tile = -1;
        if (small_tile < 0) { const char *e = getenv("PB_TILE_SMALL"); small_tile = e ? atoi(e) : TILE_128; }
        if (tile == TILE_128 && (epi == EPI_STD || epi == EPI_F32) && small_tile != TILE_128) tile = small_tile;
    }
    // split-K (gemm.h splitk): only the generic 128 x 128 tile, only when the caller lent a workspace, only launches that leave CUs idle
    a.splitk = 1;
    if (tile == TILE_128 && a.sk_ws && (epi == EPI_STD || epi == EPI_RESID)) {
        static int sk_max = -1, ncu = 0;
        if (sk_max < 0) {
            sk_max = pb_env_int("PB_SPLITK", 8);           // largest factor; 0 / 1 = off
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
        }
        const int64_t tiles = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
        const int nk = a.K / 64;
        if (sk_max > 1 && tiles < ncu) {
            int S = (int)std::min<int64_t>(sk_max, 2 * (int64_t)ncu / tiles);      // two workgroups of this tile fit a CU
            S = std::min(S, nk / 8);                                              // a slice keeps >= 8 K tiles: its prologue / reduction must stay small beside it
            while (S > 1 && (int64_t)S * tiles * 128 * 128 > a.sk_cap) --S;
            if (S > 1) a.splitk = S;
        }
    }
    if (tile == TILE_128x96 && a.Wcw && g_n96 >= 2 && !a.cTapInner) {        // packed-channel K axis (gemm.h Wcw)
        a.W = a.Wcw; a.K = a.Kcw; a.nk16 = a.nk16cw; a.mx_period = 0; a.kwrap = 0; a.kshift = 0; a.cwalk = 1;
    }
    if (tile == TILE_128x96) PB_CHECK(a.K / 64 <= 128, -1, "conv: the 128 x 96 tile's chunk table holds 128 K tiles (K = %d)", a.K);
    if (tile == TILE_128x96)
        PB_CHECK(amode == A_CONV && epi == EPI_STD && a.N <= 96 && (a.act == ACT_NONE || a.act == ACT_RELU) && !a.out2 && !a.o8_off, -1,
                 "gemm: the 128 x 96 tile serves convolutions with N <= 96, a linear / ReLU output and no copies (N = %d, act %d)", a.N, a.act);
    if (epi == EPI_RESID) PB_CHECK(a.resid && (int64_t)a.M * a.ldr * 4 < (1LL << 32) - (1 << 20), -1, "residual epilogue: the stream (%d rows) must fit a 32-bit buffer resource", a.M);
    if (epi == EPI_QKV) PB_CHECK(a.D % (tile == TILE_128 ? 128 : 256) == 0 && a.ntp % 8 == 0, -1, "qkv epilogue: D=%d ntp=%d", a.D, a.ntp);
    {   // PB_EPI_REPORT=1 (diagnostic): EPI_STD launches whose activation / skip / copy combination is not one of the
        // straight-line epilogue copies (gemm_kernels.h direct_epilogue_any PB_FAST_CASE list; tests/test_build_checks_cpu.py holds the two
        // lists against each other) take the run-time switched pass loop at ~2x the epilogue cost: name them, once per combination
        static const int report = pb_env_int("PB_EPI_REPORT", 0);
        if (report && epi == EPI_STD) {
            static const int fast_keys[] = {ACT_NONE, ACT_RELU, ACT_GELU, ACT_GRU_ZR, ACT_GRU_Q, ACT_NONE | 64, ACT_NONE | 16, ACT_NONE | 16 | 32 | 64,
                                            ACT_GRU_ZR | 16 | 32, ACT_GRU_Q | 16 | 32, ACT_RELU | 16 | 128, ACT_RELU | 64, ACT_RELU | 16};
            const int key = a.act | (a.add1 ? 16 : 0) | (a.add2 ? 32 : 0) | (a.out2 ? 64 : 0) | (a.pre_relu ? 128 : 0);
            bool hit = false;
            for (int k : fast_keys) hit |= k == key;
            static unsigned long long seen[8] = {};
            if (!hit && !(seen[key >> 6 & 7] >> (key & 63) & 1)) {
                seen[key >> 6 & 7] |= 1ull << (key & 63);
                fprintf(stderr, "[pb_epi_report] EPI_STD launch outside the fast-epilogue list: key %d (act %d%s%s%s%s), first seen at M = %d, N = %d, K = %d, %s\n", key,
                        a.act, a.add1 ? " + add1" : "", a.add2 ? " + add2" : "", a.out2 ? " + out2" : "", a.pre_relu ? " + pre_relu" : "", a.M, a.N, a.K,
                        amode == A_CONV ? "conv" : "dense");
            }
        }
    }
    const bool mx = a.nk16 > 0;
    // MX builds write / read split maps with e4m3 residual parts, fp16-only builds with fp16 residual parts
    PB_CHECK(!a.lo_off || (a.lo8 != 0) == mx, -1, "gemm: split-map format (lo8 = %d) does not match the weights' (nk16 = %d)", a.lo8, a.nk16);
    PB_CHECK(!a.o8_off || mx, -1, "gemm: an fp8 output copy needs an MX build");
    if (amode == A_DENSE && epi == EPI_STD) return mx ? pb_gemm_dense_std_mx(stream, tile, a) : pb_gemm_dense_std_f16(stream, tile, a);
    if (amode == A_DENSE && epi == EPI_RESID) return mx ? pb_gemm_dense_resid_mx(stream, tile, a) : pb_gemm_dense_resid_f16(stream, tile, a);
    if (amode == A_DENSE && epi == EPI_QKV) return mx ? pb_gemm_dense_qkv_mx(stream, tile, a) : pb_gemm_dense_qkv_f16(stream, tile, a);
    if (amode == A_DENSE && epi == EPI_PIXSHUF) return mx ? pb_gemm_dense_pixshuf_mx(stream, tile, a) : pb_gemm_dense_pixshuf_f16(stream, tile, a);
    if (amode == A_DENSE && epi == EPI_PATCH) { PB_CHECK(!mx, -1, "gemm: no MX build of epilogue %d", epi); return pb_gemm_dense_patch_f16(stream, tile, a); }
    if (amode == A_DENSE && epi == EPI_F32) { PB_CHECK(!mx, -1, "gemm: no MX build of epilogue %d", epi); return pb_gemm_dense_f32_f16(stream, tile, a); }
    if (amode == A_CONV && epi == EPI_F32) { PB_CHECK(!mx, -1, "gemm: no MX build of epilogue %d", epi); return pb_gemm_conv_f32_f16(stream, tile, a); }
    if (amode == A_CONV && epi == EPI_STD) return mx ? pb_gemm_conv_std_mx(stream, tile, a) : pb_gemm_conv_std_f16(stream, tile, a);
    if (amode == A_CONV && epi == EPI_PIXSHUF) return mx ? pb_gemm_conv_pixshuf_mx(stream, tile, a) : pb_gemm_conv_pixshuf_f16(stream, tile, a);
    if (amode == A_CONV && epi == EPI_HEAD) return mx ? pb_gemm_conv_head_mx(stream, tile, a) : pb_gemm_conv_head_f16(stream, tile, a);
    PB_CHECK(false, -1, "gemm: unsupported amode/epilogue %d/%d", amode, epi);
}
