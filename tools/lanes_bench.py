"""Experiment: split a batch over L pb_ctx (own stream + arena each) so one lane's tail wave / epilogues overlap the
other's main loops.  python tools/lanes_bench.py [batch] [lanes] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prisma_amd import engine, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
H, W = 1080, 1920
cfg = synth.DEPTH_CFGS["vitl"]
w = synth.depth_anything_weights("vitl", seed=1234)
per = B // L
nets = [engine.DepthAnything(w, "vitl", max_batch=per) for _ in range(L)]
frames = torch.from_numpy(synth.frames(B, H, W, seed=100)).cuda()
rgb = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda")
mm = torch.empty((2, B), dtype=torch.float32, device="cuda")
def step():
    for l, n in enumerate(nets):
        o = l * per
        n.infer_dev(frames[o:].data_ptr(), per, H, W, 0, rgb[o:].data_ptr(), mm[0, o:].data_ptr(), mm[1, o:].data_ptr(), True)
    for n in nets: n.sync()
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print("batch %d lanes %d: %.1f frames/s  %.2f ms/step" % (B, L, B / dt, dt * 1e3))
