"""Three-band pipeline over a 32-frame 1080p clip: how should the bands share the GPU?
python tools/pipeline_order_bench.py   -> frames/s for (a) one after the other, (b) depth, then flow || mask, (c) all three at once"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prisma_amd import engine, synth
H, W, B = 1080, 1920, 32
dn = engine.DepthAnything(synth.depth_anything_weights("vitl", seed=1234), "vitl", max_batch=B)
fn = engine.FlowRaft(synth.raft_weights(seed=4321))
mcfg = synth.MASK_CFGS["r101"]
mn = engine.MaskMMDet(synth.solov2_weights(mcfg), mcfg, max_batch=32)
frames = torch.from_numpy(synth.frame_pair_sequence(B, H, W, seed=90)).cuda()
d_rgb = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda"); d_mm = torch.empty((2, B), dtype=torch.float32, device="cuda")
sh, sw = engine.flow_out_size(H, W, 0.75)
f_rgb = torch.empty((B - 1, sh, sw, 3), dtype=torch.uint8, device="cuda"); f_mx = torch.empty((B - 1,), dtype=torch.float32, device="cuda")
m_out = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda")
keep = [synth.COCO_CLASSES.index(c) for c in synth.BAND_CLASSES]
D = lambda: dn.infer_dev(frames.data_ptr(), B, H, W, 0, d_rgb.data_ptr(), d_mm[0].data_ptr(), d_mm[1].data_ptr(), True)
F = lambda: fn.infer_sequence_dev(frames.data_ptr(), B, H, W, 0.75, 12, False, 0, f_rgb.data_ptr(), f_mx.data_ptr())
M = lambda: mn.infer_batch_dev(frames.data_ptr(), B, H, W, 0.5, keep, m_out.data_ptr())
def a(): D(); dn.sync(); F(); fn.sync(); M(); mn.sync()
def b(): D(); dn.sync(); F(); M(); fn.sync(); mn.sync()
def c(): D(); F(); M(); dn.sync(); fn.sync(); mn.sync()
def d(): D(); F(); dn.sync(); fn.sync()
def e(): D(); dn.sync(); F(); fn.sync()
for name, fnc in (("sequential", a), ("depth, then flow || mask", b), ("all three at once", c), ("depth || flow", d), ("depth, flow", e)):
    fnc(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): fnc()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print(f"{name:28s} {B / dt:7.1f} frames/s  {dt * 1e3:7.1f} ms per {B} frames", flush=True)
