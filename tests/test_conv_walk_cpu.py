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
