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

// ---- chunk walk (the 128 x 96 tile: gemm_kernels.h CW builds): one table word PER 16-BYTE CHUNK of a K tile's 128-byte row (8 per tile; a lane always
// stages the same chunk index, so it reads its own word), which lets a K tile take its chunks from several taps.  Word format as above, the chunk's own
// offset inside the pixel included; kNoChunk = a chunk of padding (the DMA reads zeros, the weights hold zeros).
constexpr unsigned kNoChunk = 0xFFFFFFFFu;
PB_CW unsigned ktab_sel_chunk(unsigned e) { return e == kNoChunk ? kNoChunk : ktab_sel(e); }       // bits no tap mask has: the tap test fails for every pixel
// classic K layouts (everything conv_ktab_word describes): the tile's word plus the chunk
PB_CW unsigned conv_ctab_classic(unsigned tile_word, int c) { return tile_word + ((unsigned)c << 6); }
// Packed-channel K axis over mx3 maps (pixel = [hi fp16 (cpad) | hi8 (cpad bytes) | lo8 (cpad bytes)], gemm.h lo8) whose cpad channels hold creal < cpad
// real ones (creal % 16 == 0: RAFT's encoder stage 2 carries 96 channels as 128): only real channels are walked.  First every tap's fp16 chunks
// (creal / 8 per tap, tap-major), padded to whole tiles - cw3_tiles16 of them - then, per tap, its hi8 chunks and its lo8 chunks (creal / 16 each), padded
// again (cw3_tiles8).  9 taps x 96 channels: 14 + 14 tiles where the per-tap layout walks 9 x (2 + 2).  q = tile * 8 + chunk.
PB_CW int cw3_tiles16(int taps, int creal) { return (taps * (creal / 8) + 7) / 8; }
PB_CW int cw3_tiles8(int taps, int creal) { return (taps * 2 * (creal / 16) + 7) / 8; }
PB_CW unsigned conv_cw3_word(int cKW, int taps, int creal, int cpad, int cW, int cld, int q) {
    const int n16 = taps * (creal / 8), q8 = cw3_tiles16(taps, creal) * 8;
    int tp, off;                                            // tap, byte offset inside the pixel
    if (q < q8) {
        if (q >= n16) return kNoChunk;
        const int per = creal / 8;
        tp = q / per; off = (q - tp * per) * 16;
    } else {
        const int per = creal / 16, r = q - q8;
        if (r >= taps * 2 * per) return kNoChunk;
        tp = r / (2 * per);
        const int rr = r - tp * 2 * per, part = rr / per;
        off = (2 + part) * cpad + (rr - part * per) * 16;
    }
    const int ky = tp / cKW, kx = tp - ky * cKW;
    const unsigned bytes = (unsigned)((ky * cW + kx) * cld * 2 + off);
    return ((bytes >> 4) << 6) | (unsigned)(ky << 3) | (unsigned)kx;
}
