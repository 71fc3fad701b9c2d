// The implicit-GEMM convolution's K walk and tap tests (gemm_kernels.h), as plain C++ so that the same text runs in the kernels and - compiled by
// g++ - in tests/test_conv_walk_cpu.py against a brute-force restatement.  Test infrastructure includes this file; it includes nothing.
#pragma once
#if defined(__HIPCC__)
#define PB_CW __device__ __host__ __forceinline__
#else
#define PB_CW inline
#endif

// which taps of a pixel lie inside the image, as bits - bit ky: input row iy0 + ky, bit 8 + kx: input column ix0 + kx (kernels up to 8 x 8).
// The K loop tests a tap with one AND + one compare against the scalar (1 << ky) | (256 << kx).
PB_CW unsigned tap_range(int v0, int n) {          // bits t of [0, 8) with 0 <= v0 + t < n
    const int lo = v0 < 0 ? -v0 : 0, hi = n - 1 - v0 < 7 ? n - 1 - v0 : 7;
    return (hi >= lo && lo < 8) ? (((2u << (hi & 7)) - 1u) & ~((1u << (lo & 7)) - 1u)) : 0u;
}
PB_CW unsigned tap_mask(int iy0, int ix0, int H, int W) { return tap_range(iy0, H) | (tap_range(ix0, W) << 8); }

// ... and the K walk of a convolution as a table in the LDS, one word per K tile, built once per workgroup: the tile's tap (ky, kx) in bits
// [0, 6) and, above them, the byte offset / 16 of (tap, channel slice) relative to the pixel's tap (0, 0).  Both K orders (gemm.h cTapInner),
// the split-fp16 wrap (kwrap / kshift) and the clamp past the last tile are in the table, so the K loop pays one broadcast ds_read, one
// v_readfirstlane and five scalar instructions per K tile instead of the ~30 scalar instructions of a branch-free cursor step PER A HALF
// (round 4, per-tile stamps: the 3 x 3 / 1 x 5 convolutions of the RAFT update block ran 3260 cycles per K tile against the dense GEMM's 2320 -
// the ping-pong schedule has no room for ~100 extra instructions per K tile in the loading wave group's slots).
// tapin: slice-major K order (c / 64, tap, c % 64) instead of tap-major (tap, c); cC = channels per tap of the concatenated K axis (64-half units),
// cld = pixel stride in halfs; a channel cursor c0 >= kwrap (kwrap != 0) reads channel c0 + kshift (gemm.h).
PB_CW unsigned conv_ktab_word(int tapin, int cKH, int cKW, int cC, int cW, int cld, int kwrap, int kshift, int t) {
    int ky, kx, c0;
    if (tapin) {
        const int per = cKH * cKW, sl = t / per, tp = t - sl * per;
        ky = tp / cKW; kx = tp - ky * cKW; c0 = sl * 64;
    } else {
        const int cpt = cC >> 6, tp = t / cpt;
        ky = tp / cKW; kx = tp - ky * cKW; c0 = (t - tp * cpt) * 64;
    }
    const int cs = (kwrap && c0 >= kwrap) ? c0 + kshift : c0;
    const unsigned bytes = (unsigned)(((ky * cW + kx) * cld + cs) * 2);
    return ((bytes >> 4) << 6) | (unsigned)(ky << 3) | (unsigned)kx;
}
PB_CW unsigned ktab_bytes(unsigned e) { return (e >> 6) << 4; }
PB_CW unsigned ktab_sel(unsigned e) { return (1u << ((e >> 3) & 7)) | (256u << (e & 7)); }
