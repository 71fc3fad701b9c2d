"""Per-tile timing breakdown of the ping-pong GEMM (PB_GEMM_DBG stamps; gemm8_kernel writes one record per TILE: start = kernel start or
the previous tile's epilogue issued, loop start, loop end, epilogue issued).  Environment switches of the kernel apply
(PB_GEMM_PERSIST, PB_GEMM_PREFETCH, PB_GEMM_ABL, PB_GEMM_STAGGER); `python tools/gemm_stamps.py [shape ...]` with shapes out of
fc1 fc1-gelu proj fc2 sq8k conv3x3 conv3x3-slice conv1x5-gru dense-k2304 (default: all)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PB_GEMM_DBG"] = "/tmp/gemm_dbg.bin"
from prisma_amd import engine
ops = engine.Ops(0)
SHAPES = {"fc1": (78336, 4096, 1024, 0), "fc1-gelu": (78336, 4096, 1024, 1), "proj": (78336, 1024, 1024, 2), "fc2": (78336, 1024, 4096, 2),
          "sq8k": (8192, 8192, 8192, 0),
          # implicit-GEMM convolutions on the RAFT update block's grid (31 pairs x 102 x 180 rows): 3 x 3 256 -> 256 tap-major / slice-major,
          # the GRU's 1 x 5 over 384 channels, and the dense GEMM of the same M, N, K beside them
          "conv3x3": (31 * 18360, 256, 2304, 10), "conv3x3-slice": (31 * 18360, 256, 2304, 11), "conv1x5-gru": (31 * 18360, 256, 1920, 12),
          "dense-k2304": (31 * 18360, 256, 2304, 0)}
want = sys.argv[1:] or list(SHAPES)
print("# switches:", {k: v for k, v in os.environ.items() if k.startswith("PB_GEMM_") and k != "PB_GEMM_DBG"})
for name in want:
    m, n, k, epi = SHAPES[name]
    ms = ops.gemm_bench(m, n, k, tile=2, epi=epi, iters=5)
    d = np.fromfile("/tmp/gemm_dbg.bin", dtype=np.int64).reshape(-1, 8)
    d = d[d[:, 3] != 0]
    pro, loop, epi_c, tot = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 3] - d[:, 0]
    real = (d[:, 5] - d[:, 4]) / 100.0     # us (100 MHz)
    flops = 2.0 * m * n * k
    print(f"{name}: {ms:.3f} ms/launch = {flops / ms * 1e-9:.0f} TF/s, tiles {len(d)}; cycles per tile (median): start->loop {np.median(pro):.0f} "
          f"loop {np.median(loop):.0f} epilogue issue {np.median(epi_c):.0f} total {np.median(tot):.0f}; p10/p90 of total {np.percentile(tot, 10):.0f}/{np.percentile(tot, 90):.0f}; "
          f"tile wall median {np.median(real):.2f} us (clock ~{np.median(tot / np.maximum(real, 1e-3)) / 1e3:.2f} GHz); loop/k-tile {np.median(loop) / (k // 64):.0f} cyc")
