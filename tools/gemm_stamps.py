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
    print(f"    epilogue: issue {np.median(d[:, 6] - d[:, 2]):.0f} cycles, then drain (vmcnt 0) {np.median(d[:, 3] - d[:, 6]):.0f}; "
          f"tile start spread (us, p10-p90 of first 256 blocks): {np.percentile((d[:256, 4] - d[:256, 4].min()) / 100.0, [10, 50, 90])}; "
          f"end-time spread of blocks 256..511: {np.percentile((d[256:512, 5] - d[256:512, 5].min()) / 100.0, [10, 50, 90])}")
    real = (d[:, 5] - d[:, 4]) / 100.0     # us (100 MHz)
    print(f"{name}: {ms:.3f} ms/launch, blocks {len(d)}; cycles median: prologue {np.median(pro):.0f} loop {np.median(loop):.0f} "
          f"epilogue {np.median(epi_c):.0f} total {np.median(tot):.0f}; block wall median {np.median(real):.2f} us "
          f"(clock ~{np.median(tot / real) / 1e3:.2f} GHz); loop/k-tile {np.median(loop) / (k // 64):.0f} cyc")
