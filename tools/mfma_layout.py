"""A numpy model of `v_mfma_f32_32x32x16_f16`'s operand and accumulator layouts on gfx950, as the kernels in prisma_amd/csrc use them -
for working out data movement on paper (well, in numpy) before spending GPU time on it.  tests/test_mfma_layout_cpu.py checks the three
layout tricks the GEMM and attention kernels rest on.

  D[32 x 32] += A[32 x 16] . B[16 x 32], one wave of 64 lanes, lane = 32 * lh + li:
    A operand (f16x8 per lane):  A[li][8 lh + j],     j = 0..7     (a row of A, half of the 16 k values)
    B operand (f16x8 per lane):  B[8 lh + j][li]      (a COLUMN of B; stored as a row of W when B = W^T)
    accumulator (f32x16 per lane): D[(r & 3) + 8 (r >> 2) + 4 lh][li],  r = 0..15
"""
import numpy as np

LANES = np.arange(64)
LI, LH = LANES % 32, LANES // 32


def a_fragment(A):
    """A [32, 16] -> per-lane fragments [64, 8]"""
    return np.stack([A[LI[l], 8 * LH[l]: 8 * LH[l] + 8] for l in LANES])


def b_fragment(Wrows):
    """B = W^T with W [32 (columns of B), 16 (k)] -> per-lane fragments [64, 8]: lane (li, lh) holds W[li][8 lh .. 8 lh + 7]"""
    return np.stack([Wrows[LI[l], 8 * LH[l]: 8 * LH[l] + 8] for l in LANES])


def acc_rows():
    """[64, 16]: the D row held by accumulator register r of each lane (the column is li)"""
    r = np.arange(16)
    return (r[None, :] & 3) + 8 * (r[None, :] >> 2) + 4 * LH[:, None]


def mfma(a_frag, b_frag, acc):
    """acc [64, 16] += the product the hardware forms from these fragments (fp32 accumulate of fp16 operands)"""
    A = np.zeros((32, 16), np.float32)
    Bt = np.zeros((32, 16), np.float32)
    for l in LANES:
        A[LI[l], 8 * LH[l]: 8 * LH[l] + 8] = a_frag[l]
        Bt[LI[l], 8 * LH[l]: 8 * LH[l] + 8] = b_frag[l]
    D = A.astype(np.float32) @ Bt.astype(np.float32).T
    rows = acc_rows()
    return acc + D[rows, LI[:, None]]


def swap_bits23(i):
    """attention.hip `kperm`: the order the K rows of a 32-key sub-tile are fed in, so that accumulator registers 8 s .. 8 s + 7 of a
    lane are the 8 consecutive keys (16 s + 8 lh ..) of the B fragment of k-step s of the second matmul"""
    i = np.asarray(i)
    return (i & 19) | ((i & 4) << 1) | ((i & 8) >> 1)


def col_map(r):
    """gemm_kernels.h: LDS row r of a 64-column weight tile holds output column 2 (r & 31) + (r >> 5): a lane's two column blocks are
    then two ADJACENT columns, stored as one dword"""
    r = np.asarray(r)
    return (r & ~63) + 2 * (r & 31) + ((r >> 5) & 1)
