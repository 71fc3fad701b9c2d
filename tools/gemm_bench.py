"""GEMM kernel micro-benchmark through the C ABI (device-resident random fp16 data)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prisma_amd import engine

ops = engine.Ops(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M = B * 2448
shapes = [("qkv", M, 3072, 1024, 0), ("proj", M, 1024, 1024, 2), ("fc1", M, 4096, 1024, 1), ("fc2", M, 1024, 4096, 2),
          ("fc1-noact", M, 4096, 1024, 0), ("sq8k", 8192, 8192, 8192, 0)]
tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4, 2]
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
for name, m, n, k, epi in shapes:
    if only and name not in only:
        continue
    row = []
    for t in tiles:
        ms = ops.gemm_bench(m, n, k, tile=t, epi=epi, iters=10)
        row.append(f"tile{t}: {ms:7.3f} ms {2.0 * m * n * k / ms / 1e9:7.1f} TF")
    print(f"{name:10s} M={m} N={n} K={k} epi={epi} | " + " | ".join(row), flush=True)
