"""Per-block timing breakdown of the ping-pong GEMM (PB_GEMM_DBG stamps)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PB_GEMM_DBG"] = "/tmp/gemm_dbg.bin"
from prisma_amd import engine
ops = engine.Ops(0)
for name, m, n, k, epi in [("fc1", 78336, 4096, 1024, 0), ("fc1-gelu", 78336, 4096, 1024, 1), ("proj", 78336, 1024, 1024, 2), ("sq8k", 8192, 8192, 8192, 0)]:
    ms = ops.gemm_bench(m, n, k, tile=2, epi=epi, iters=3)
    d = np.fromfile("/tmp/gemm_dbg.bin", dtype=np.int64).reshape(-1, 8)
    d = d[d[:, 3] != 0]
    pro, loop, epi_c, tot = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 3] - d[:, 0]
    real = (d[:, 5] - d[:, 4]) / 100.0     # us (100 MHz)
    print(f"{name}: {ms:.3f} ms/launch, blocks {len(d)}; cycles median: prologue {np.median(pro):.0f} loop {np.median(loop):.0f} "
          f"epilogue {np.median(epi_c):.0f} total {np.median(tot):.0f}; block wall median {np.median(real):.2f} us "
          f"(clock ~{np.median(tot / real) / 1e3:.2f} GHz); loop/k-tile {np.median(loop) / (k // 64):.0f} cyc")
    # gap between consecutive blocks on the same CU
    hw = d[:, 6]
    cu = (hw >> 8) & 0xffffff   # everything above wave/simd id
    gaps = []
    for c in np.unique(cu):
        r = d[cu == c]
        r = r[np.argsort(r[:, 4])]
        gaps += list((r[1:, 4] - r[:-1, 5]) / 100.0)
    if gaps:
        print(f"    per-CU gap between consecutive blocks: median {np.median(gaps):.2f} us, p90 {np.percentile(gaps, 90):.2f} us; "
              f"span {(d[:, 5].max() - d[:, 4].min()) / 100.0:.1f} us")
