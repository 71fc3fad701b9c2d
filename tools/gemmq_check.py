"""Correctness + speed of the quad GEMM kernel (tile 6) against the ping-pong kernel (tile 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prisma_amd import engine
ops = engine.Ops(0)
rng = np.random.default_rng(0)
for (M, N, K) in [(512, 512, 256), (700, 768, 320), (1000, 256, 64), (300, 1024, 1024)]:
    A = rng.standard_normal((M, K)).astype(np.float32); W = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = A.astype(np.float16).astype(np.float32) @ W.astype(np.float16).astype(np.float32).T + b
    for tile in (2, 6):
        got = ops.gemm(A, W, b, act=0, tile=tile)
        err = np.abs(got - ref).max() / np.abs(ref).max()
        print(f"M={M} N={N} K={K} tile={tile} relmax {err:.2e}", flush=True)
B = 32; M = B * 2448
for name, m, n, k, epi in [("qkv", M, 3072, 1024, 0), ("fc1", M, 4096, 1024, 1), ("fc2", M, 1024, 4096, 2), ("proj", M, 1024, 1024, 2), ("sq8k", 8192, 8192, 8192, 0)]:
    row = []
    for t in (2, 6):
        ms = ops.gemm_bench(m, n, k, tile=t, epi=epi, iters=10)
        row.append(f"tile{t}: {ms:7.3f} ms {2.0 * m * n * k / ms / 1e9:7.1f} TF")
    print(f"{name:6s} M={m} N={n} K={k} epi={epi} | " + " | ".join(row), flush=True)
