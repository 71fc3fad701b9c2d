"""What bounds the K = 256 correlation-volume GEMM?  Same tile count and output bytes, different output shapes:
python tools/volume_gemm_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prisma_amd import engine
ops = engine.Ops(0)
for name, m, n, k, tile in (("level 0 as launched: 18360 x 19136, rows 38 KB apart", 18360, 19136, 256, 1),
                            ("same, 256 x 256 ping-pong kernel (N padded to 19200)", 18360, 19200, 256, 2),
                            ("same tiles, output contiguous per tile: 2754000 x 128", 2754000, 128, 256, 1),
                            ("same tiles, N = 256 (512 B per row)", 1377000, 256, 256, 1),
                            ("K = 1024 for scale: 18360 x 19136", 18360, 19136, 1024, 1)):
    ms = ops.gemm_bench(m, n, k, tile=tile, epi=0, iters=10)
    print(f"{name:62s} {ms:7.3f} ms  out {2.0 * m * n / ms / 1e9:7.1f} GB/s  {2.0 * m * n * k / ms / 1e9:7.1f} TF/s", flush=True)
