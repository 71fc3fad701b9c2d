"""256 x 128 ping-pong GEMM (tile code 10) against torch on a few conv shapes: python tools/g8n_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from prisma_amd import engine
ops = engine.Ops(0)
h = lambda a: a.astype(np.float16).astype(np.float32)
g = np.random.default_rng(3)
for tile in (10, 1):
    ops.set_option("conv_tile", tile)
    for (B, Ci, H, W, Co, ks, stride, relu) in [(1, 128, 24, 40, 128, 3, 1, 0), (2, 256, 33, 21, 128, 3, 1, 1), (1, 64, 40, 56, 192, 3, 2, 0),
                                               (3, 384, 17, 23, 128, 1, 1, 1), (1, 128, 90, 160, 128, 3, 1, 0)]:
        x = h(g.standard_normal((B, Ci, H, W))); w = h(g.standard_normal((Co, Ci, ks, ks)) / np.sqrt(Ci * ks * ks)); b = g.standard_normal(Co).astype(np.float32)
        ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=stride, padding=ks // 2)
        if relu: ref = ref.relu()
        out = ops.conv2d(x, w, b, stride=stride, relu_out=bool(relu))
        err = float(np.abs(out - ref.numpy()).max() / np.abs(ref.numpy()).max())
        print(f"tile {tile:2d} B{B} Ci{Ci} {H}x{W} Co{Co} k{ks} s{stride} relu{relu}: relmax {err:.2e}", flush=True)
