"""CPU: the address arithmetic of the buffer-addressed pixel-shuffle epilogue (prisma_amd/csrc/pixshuf_walk.h - the very text the GEMM kernels
compile) built with g++ and held against the flat epilogue's (b, y, x) formula: for every row of a wave tile, first pixel of the wave +
s x (tile-local row) + wraps x s (s - 1) ps_w must be the output pixel ((b ps_h + y) s + dy) (ps_w s) + x s + dx - over grid widths from the
launcher's minimum (32) to wider than a tile, both wave-tile heights (64 and 128 rows), shuffles of 2 and 4, first rows anywhere in the grid."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include <cstdlib>
#include "pixshuf_walk.h"
int main(int argc, char **argv) {       // TM s psw dy dx m0...: prints first pixel, then the wraps of every tile-local row
    const int TM = atoi(argv[1]), s = atoi(argv[2]), psw = atoi(argv[3]), dy = atoi(argv[4]), dx = atoi(argv[5]);
    for (int a = 6; a < argc; ++a) {
        const int m0 = atoi(argv[a]), x0 = m0 - (m0 / psw) * psw;
        printf("%lld", pixshuf_first_pixel(m0, s, psw, dy, dx));
        for (int d = 0; d < TM * 32; ++d) printf(" %d", TM == 4 ? pixshuf_wraps<4>(x0 + d, psw) : pixshuf_wraps<2>(x0 + d, psw));
        printf("\n");
    }
    return 0;
}
"""


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("pixshuf_walk")
    src = d / "walk.cpp"
    src.write_text(SRC)
    out = d / "walk"
    subprocess.run(["g++", "-O1", "-I", os.path.join(ROOT, "prisma_amd", "csrc"), str(src), "-o", str(out)], check=True)
    return str(out)


def flat_pixel(m, s, psh, psw, dy, dx):
    """direct_epilogue_f16_impl's EPI_PIXSHUF offset / ldo: two divisions per row."""
    hw = psh * psw
    b, rem = divmod(m, hw)
    y, x = divmod(rem, psw)
    return ((b * psh * s + (y * s + dy)) * (psw * s)) + (x * s + dx)


@pytest.mark.parametrize("TM", [2, 4])
@pytest.mark.parametrize("s,dy,dx", [(2, 0, 0), (2, 1, 1), (4, 3, 2), (1, 0, 0)])
@pytest.mark.parametrize("psw,psh", [(32, 5), (33, 19), (46, 34), (64, 7), (77, 43), (127, 3), (128, 9), (360, 203), (1000, 2)])
def test_wave_pixel_walk_equals_flat_formula(exe, TM, s, dy, dx, psw, psh):
    rng = np.random.default_rng(psw * 31 + s)
    total = 3 * psh * psw                                       # three images: rows run on across the batch
    m0s = sorted(set([0, psw - 1, psw, psh * psw - 1, psh * psw, total - TM * 32] + [int(v) for v in rng.integers(0, total - TM * 32, 12)]))
    m0s = [m for m in m0s if 0 <= m <= total - TM * 32]
    r = subprocess.run([exe, str(TM), str(s), str(psw), str(dy), str(dx)] + [str(m) for m in m0s], capture_output=True, text=True, check=True)
    lines = r.stdout.strip().split("\n")
    assert len(lines) == len(m0s)
    for m0, line in zip(m0s, lines):
        v = [int(t) for t in line.split()]
        pix0, wraps = v[0], v[1:]
        assert pix0 == flat_pixel(m0, s, psh, psw, dy, dx), (m0, psw, s)
        for d, w in enumerate(wraps):
            assert pix0 + s * d + w * s * (s - 1) * psw == flat_pixel(m0 + d, s, psh, psw, dy, dx), (m0, d, psw, s, TM)
