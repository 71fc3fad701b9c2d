"""Attention kernel microbenchmark (ViT-L shape of the bench: B=32, 16 heads, 2443 tokens).
python tools/attn_bench.py [variant ...]      (variants: see launch_attention in attention.hip; everything but 0 and 2 needs a -DPB_DIAG build of the library:
cd prisma_amd/csrc && make EXTRA=-DPB_DIAG BUILD=build_diag LIB=../libprisma_bands_diag.so; PRISMA_BANDS_LIB=prisma_amd/libprisma_bands_diag.so python tools/attn_bench.py 0 6 7)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prisma_amd import engine

B, heads, N = int(os.environ.get("ATT_B", 32)), 16, int(os.environ.get("ATT_N", 2443))
net = engine.Ops(0)
flop = 4.0 * N * N * 64 * heads * B
for v in [int(a) for a in sys.argv[1:]] or [0]:
    ms = net.attention_bench(B, heads, N, v, int(os.environ.get("ATT_ITERS", 10)))
    print(f"variant {v}: {ms:.4f} ms  {flop / ms / 1e9:.1f} TF", flush=True)
