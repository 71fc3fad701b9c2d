// Address arithmetic of the buffer-addressed pixel-shuffle epilogue (gemm_kernels.h pixshuf_epilogue_buf), as plain C++ so that the same text
// runs in the kernels and - compiled by g++ - in tests/test_pixshuf_walk_cpu.py against the (b, y, x) formula of the flat epilogue.
// Test infrastructure includes this file; it includes nothing.
#pragma once
#if defined(__HIPCC__)
#define PB_PW __device__ __host__ __forceinline__
#else
#define PB_PW inline
#endif

// Row m = (b, y, x) of a [B, ps_h, ps_w] grid, tap (dy, dx) of an s x s shuffle: output pixel ((b ps_h + y) s + dy) (ps_w s) + x s + dx.
// With Y = m / ps_w (grid rows run on across the batch) that is s m + s (s - 1) ps_w Y + dy ps_w s + dx: one division.
PB_PW long long pixshuf_first_pixel(int m, int s, int psw, int tap_dy, int tap_dx) {
    const int Y0 = m / psw;
    return (long long)s * m + (long long)s * (s - 1) * psw * Y0 + (long long)tap_dy * psw * s + tap_dx;
}
// ... and row m + d lies wraps(x0 + d) grid rows further down, x0 = m % ps_w: each adds s (s - 1) ps_w pixels to the s d of the row itself.
// d < 32 TM and ps_w >= 32 (the launcher's condition): at most TM wraps.
template <int TM>
PB_PW int pixshuf_wraps(int t, int psw) {
    int w = 0;
#pragma unroll
    for (int k = 1; k <= TM; ++k) w += t >= k * psw ? 1 : 0;
    return w;
}
