"""CPU: the implicit-GEMM convolution's K walk and tap tests (prisma_amd/csrc/conv_walk.h - the very text the GEMM kernels compile) built with g++
and held against brute force: the table word of every K tile against a step-by-step cursor walk in both K orders with and without the split-fp16
wrap (gemm.h kwrap / kshift), and the 8 + 8 bit tap mask against per-tap range tests for every first row / column a padded convolution can have."""
import itertools
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <cstdlib>
#include "conv_walk.h"
int main(int argc, char **argv) {
    if (argv[1][0] == 't') {            // t tapin KH KW cC cW cld kwrap kshift nk
        int a[9];
        for (int i = 0; i < 9; ++i) a[i] = atoi(argv[2 + i]);
        for (int t = 0; t < a[8]; ++t) {
            const unsigned e = conv_ktab_word(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], t);
            printf("%u %u\n", ktab_bytes(e), ktab_sel(e));
        }
    } else if (argv[1][0] == 'c') {     // c KW taps creal cpad cW cld: the packed-channel chunk words of every K tile (conv_cw3_word)
        int a[6];
        for (int i = 0; i < 6; ++i) a[i] = atoi(argv[2 + i]);
        const int nq = (cw3_tiles16(a[1], a[2]) + cw3_tiles8(a[1], a[2])) * 8;
        printf("%d %d\n", cw3_tiles16(a[1], a[2]), cw3_tiles8(a[1], a[2]));
        for (int q = 0; q < nq; ++q) {
            const unsigned e = conv_cw3_word(a[0], a[1], a[2], a[3], a[4], a[5], q);
            printf("%u %u\n", e == kNoChunk ? 0u : ktab_bytes(e), ktab_sel_chunk(e));
        }
    } else if (argv[1][0] == 'x') {     // x <the arguments of t>: classic layouts expanded per chunk (conv_ctab_classic)
        int a[9];
        for (int i = 0; i < 9; ++i) a[i] = atoi(argv[2 + i]);
        for (int q = 0; q < a[8] * 8; ++q) {
            const unsigned e = conv_ctab_classic(conv_ktab_word(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], q >> 3), q & 7);
            printf("%u %u\n", ktab_bytes(e), ktab_sel_chunk(e));
        }
    } else {                            // m H W lo hi: masks for every (iy0, ix0) in [lo, hi)
        const int H = atoi(argv[2]), W = atoi(argv[3]), lo = atoi(argv[4]), hi = atoi(argv[5]);
        for (int y = lo; y < hi; ++y)
            for (int x = lo; x < hi; ++x) printf("%u\n", tap_mask(y, x, H, W));
    }
    return 0;
}
"""


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("conv_walk")
    src = d / "walk.cpp"
    src.write_text(SRC)
    out = d / "walk"
    subprocess.run(["g++", "-O1", "-I", os.path.join(ROOT, "prisma_amd", "csrc"), str(src), "-o", str(out)], check=True)
    return str(out)


def cursor_walk(tapin, KH, KW, cC, cW, cld, kwrap, kshift, nk):
    """the round-1..3 K loop: a (ky, kx, c0) cursor stepped once per K tile (tap-major: channels innermost; slice-major: taps innermost)"""
    ky = kx = c0 = 0
    out = []
    for _ in range(nk):
        cs = c0 + kshift if (kwrap and c0 >= kwrap) else c0
        out.append((((ky * cW + kx) * cld + cs) * 2, (1 << ky) | (256 << kx)))
        if tapin:
            kx += 1
            if kx == KW:
                kx, ky = 0, ky + 1
                if ky == KH:
                    ky, c0 = 0, c0 + 64
        else:
            c0 += 64
            if c0 >= cC:
                c0, kx = 0, kx + 1
                if kx == KW:
                    kx, ky = 0, ky + 1
    return out


@pytest.mark.parametrize("tapin", [0, 1])
@pytest.mark.parametrize("KH,KW", [(3, 3), (1, 5), (5, 1), (1, 1), (7, 7)])
@pytest.mark.parametrize("C,split", [(64, 0), (256, 0), (128, 1), (256, 2), (256, 3)])
def test_table_equals_cursor_walk(exe, tapin, KH, KW, C, split):
    # split 0: plain fp16 map; 1: [hi | lo] fp16 pair, the lo pass re-reads hi (kwrap = C, kshift = -C, cC = 2 C); 2: [a16 (C) | a8 (C bytes)] (cC = 1.5 C);
    # 3: the same on the first 2/3-wide slice of a 1.5 x wider pixel, whose fp8 copy starts C / 2 halfs after the slice ends (kwrap = C, kshift > 0)
    cW = 180
    cC, cld, kwrap, kshift = {0: (C, C, 0, 0), 1: (2 * C, C, C, -C), 2: (C + C // 2, C + C // 2, 0, 0), 3: (C + C // 2, 576, C, 128)}[split]
    nk = KH * KW * cC // 64
    got = subprocess.run([exe, "t"] + [str(v) for v in (tapin, KH, KW, cC, cW, cld, kwrap, kshift, nk)], capture_output=True, text=True, check=True).stdout.split()
    got = list(zip(map(int, got[0::2]), map(int, got[1::2])))
    assert got == cursor_walk(tapin, KH, KW, cC, cW, cld, kwrap, kshift, nk)


@pytest.mark.parametrize("H,W", [(102, 180), (5, 3), (1, 1), (8, 9)])
def test_tap_mask_equals_range_tests(exe, H, W):
    lo, hi = -9, max(H, W) + 3
    got = list(map(int, subprocess.run([exe, "m", str(H), str(W), str(lo), str(hi)], capture_output=True, text=True, check=True).stdout.split()))
    want = []
    for y, x in itertools.product(range(lo, hi), repeat=2):
        m = 0
        for t in range(8):
            m |= (1 << t) if 0 <= y + t < H else 0
            m |= (256 << t) if 0 <= x + t < W else 0
        want.append(m)
    assert got == want


@pytest.mark.parametrize("KH,KW,creal,cpad", [(3, 3, 96, 128), (3, 3, 32, 64), (1, 5, 48, 64), (3, 3, 16, 64), (3, 3, 112, 128)])
def test_packed_channel_chunk_walk_lists_every_real_chunk_once(exe, KH, KW, creal, cpad):
    """conv_cw3_word (the 128 x 96 tile's packed-channel K axis over [hi fp16 (cpad) | hi8 | lo8] pixels): the fp16 region holds every (tap, 8-channel
    chunk) of the real channels exactly once, tap-major, the fp8 region every (tap, part, 16-channel chunk); both are padded to whole tiles with chunks
    no pixel can pass the tap test of; nothing points into the padding channels; and the weights' packer (engine_base.hip pack_conv) lays a row out in
    exactly this order (fp16: row[tap * creal + c]; fp8: row8[(tap * 2 + part) * creal + c])."""
    cW, cld, taps = 180, 2 * cpad, KH * KW
    got = subprocess.run([exe, "c"] + [str(v) for v in (KW, taps, creal, cpad, cW, cld)], capture_output=True, text=True, check=True).stdout.split()
    nk16, nk8 = int(got[0]), int(got[1])
    words = list(zip(map(int, got[2::2]), map(int, got[3::2])))
    assert nk16 == -(-taps * creal // 8 // 8) and nk8 == -(-taps * 2 * creal // 16 // 8) and len(words) == (nk16 + nk8) * 8
    NONE = 0xFFFFFFFF
    want16 = [(((tp // KW) * cW + tp % KW) * cld * 2 + c * 16, (1 << (tp // KW)) | (256 << (tp % KW))) for tp in range(taps) for c in range(creal // 8)]
    want8 = [(((tp // KW) * cW + tp % KW) * cld * 2 + (2 + part) * cpad + c * 16, (1 << (tp // KW)) | (256 << (tp % KW)))
             for tp in range(taps) for part in range(2) for c in range(creal // 16)]
    assert words[:len(want16)] == want16 and all(w == (0, NONE) for w in words[len(want16):nk16 * 8])
    assert words[nk16 * 8:nk16 * 8 + len(want8)] == want8 and all(w == (0, NONE) for w in words[nk16 * 8 + len(want8):])


def test_classic_layouts_expand_into_the_chunk_table(exe):
    """a classic K tile's eight chunk words are its tile word + 16 bytes per chunk, same tap"""
    args = (0, 3, 3, 256, 180, 256, 0, 0, 36)
    tile = subprocess.run([exe, "t"] + [str(v) for v in args], capture_output=True, text=True, check=True).stdout.split()
    tile = list(zip(map(int, tile[0::2]), map(int, tile[1::2])))
    got = subprocess.run([exe, "x"] + [str(v) for v in args], capture_output=True, text=True, check=True).stdout.split()
    got = list(zip(map(int, got[0::2]), map(int, got[1::2])))
    assert got == [(b + 16 * c, sel) for (b, sel) in tile for c in range(8)]
