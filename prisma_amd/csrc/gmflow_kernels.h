// Launchers of the flow_gmflow band's non-GEMM kernels (gmflow_kernels.hip).
#pragma once
#include "common.h"
#include "../../include/prisma_bands.h"

// 1/8-resolution token grid of one frame and its 2 x 2 attention windows (attn_splits = 2: bands/flow_gmflow.py:237)
struct GmGeom {
    int h8, w8, P;          // grid, tokens per frame
    int wh, ww, Lw;         // window size, tokens per window
    int ldv;                // row stride of a window's V^T (round_up(Lw, 32))
};
struct GmPackJob {
    const float *src;       // fp32 projection matrix [frames * P, ld]
    int ld, col;            // ... and the first of the 128 columns to take
    f16 *dst;               // rows: [Bw, Lw, 256] = [hi | lo];  vt: [Bw, 2, 128, ldv]
    int is_vt;
};
struct GmPackJobs { GmPackJob j[5]; int n; };

int launch_gm_tokens(hipStream_t s, const float *feat, const float *pos, float *X, f16 *Xs, int NP, int P);
int launch_gm_split_rows(hipStream_t s, const float *src, int ld, int C, f16 *dst, int64_t rows);
int launch_gm_pack(hipStream_t s, const GmPackJobs &jobs, const GmGeom &g, int Bw, int shifted);
int launch_gm_ln(hipStream_t s, const float *M, const float *gamma, const float *beta, float *X, f16 *out, int64_t rows, const GmGeom &g,
                 int windowed, int shifted, int mode);
int launch_gm_grid_vt(hipStream_t s, f16 *vt, int P, int w8, int ldv);
int launch_gm_match_flow(hipStream_t s, const float *O, float *flow, f16 *vt, int B, int P, int w8, int ldv);
int launch_gm_upsampler_in(hipStream_t s, const float *O, const float *X, float *flow, f16 *map, int B, int P, int img_step);
