"""N <= 128 layers: the generic 128 x 128 tile (1) against the 256 x 128 ping-pong kernel (11, gemm_n128.h) on the shapes the bands
launch - device-resident random fp16 operands through pb_op_gemm_bench - plus per-tile stamps of the new kernel.
python tools/n128_bench.py [shape ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prisma_amd import engine
P = 31 * 18360        # rows of the RAFT update block at 1080p x 0.75 (31 pairs x 102 x 180)
SHAPES = {"gru-q 1x5 256->128": (P, 128, 1280, 12), "motion 3x3 256->128": (P, 128, 2304, 10), "enc 3x3 128->128": (P, 128, 1152, 10),
          "enc 3x3 128->128 slice": (P, 128, 1152, 11), "dense K=1280": (P, 128, 1280, 0), "dense K=256 (1x1)": (P, 128, 256, 0),
          "enc 3x3 128->96": (P, 96, 1152, 10)}
want = sys.argv[1:] or list(SHAPES)
ops = engine.Ops(0)
for name in want:
    m, n, k, epi = SHAPES[name]
    row = []
    for t in (1, 11, 2):
        os.environ.pop("PB_GEMM_DBG", None)
        ms = ops.gemm_bench(m, n, k, tile=t, epi=epi, iters=5)
        row.append(f"tile {t}: {ms:.3f} ms {2.0 * m * n * k / ms * 1e-9:.0f} TF/s")
    os.environ["PB_GEMM_DBG"] = "/tmp/gemm_dbg.bin"
    ops.gemm_bench(m, n, k, tile=11, epi=epi, iters=2)
    os.environ.pop("PB_GEMM_DBG", None)
    d = np.fromfile("/tmp/gemm_dbg.bin", dtype=np.int64).reshape(-1, 8)
    d = d[d[:, 3] != 0]
    pro, loop, epi_c, tot = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 3] - d[:, 0]
    real = (d[:, 5] - d[:, 4]) / 100.0
    print(f"{name}: " + " | ".join(row) + f" || tile 11 stamps ({len(d)} tiles): start->loop {np.median(pro):.0f} loop {np.median(loop):.0f} "
          f"epilogue {np.median(epi_c):.0f} total {np.median(tot):.0f} cycles, loop / K tile {np.median(loop) / (k // 64):.0f} (MFMA bound 1536), "
          f"clock ~{np.median(tot / np.maximum(real, 1e-3)) / 1e3:.2f} GHz", flush=True)
