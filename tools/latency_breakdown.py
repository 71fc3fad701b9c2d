"""BASELINE configs[1]: one 1280x720 frame, batch 1, through depth_anything ViT-L - where the milliseconds go (per-family HIP-event times of the
engine's own launch records) next to the wall time of a call.  python tools/latency_breakdown.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prisma_amd import engine, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = synth.DEPTH_CFGS["vitl"]
net = engine.DepthAnything(synth.cached_weights("depth", cfg, 1234), cfg, device=0, max_batch=n)
f = torch.from_numpy(synth.frames(n, 720, 1280, seed=7)).cuda()
rgb = torch.empty((n, 720, 1280, 3), dtype=torch.uint8, device="cuda")
sc = torch.zeros((2, n), dtype=torch.float32, device="cuda")
def call():
    net.infer_dev(f.data_ptr(), n, 720, 1280, 0, rgb.data_ptr(), sc[0].data_ptr(), sc[1].data_ptr(), True)
    net.sync()
for _ in range(3):
    call()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    call()
wall = (time.perf_counter() - t0) / 10 * 1e3
net.set_profiling(timing=True)
call()
st = net.kernel_stats()
tot = sum(s["ms"] for s in st)
print(f"batch {n} 1280x720: wall {wall:.3f} ms per call, {tot:.3f} ms in {sum(s['launches'] for s in st)} launches ({wall - tot:.3f} ms between them)")
for s in sorted(st, key=lambda s: -s["ms"]):
    tf = s["flops"] / (s["ms"] * 1e-3) / 1e12 if s["ms"] > 0 and s["flops"] > 0 else 0
    print(f"  {s['name']:72s} {s['ms']:7.3f} ms  {s['launches']:4d} launches  {tf:6.0f} TF/s")
