"""Experiment: L FlowRaft contexts (own stream + arena) each taking pairs/L consecutive pairs: python tools/raft_lanes_bench.py [pairs] [lanes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prisma_amd import engine, synth
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
per = pairs // L
nets = [engine.FlowRaft(synth.raft_weights(seed=4321)) for _ in range(L)]
fr = torch.from_numpy(synth.frame_pair_sequence(pairs + 1, 720, 1280, seed=50)).cuda()
rgb = torch.empty((pairs, 720, 1280, 3), dtype=torch.uint8, device="cuda"); mx = torch.empty((pairs,), dtype=torch.float32, device="cuda")
def step():
    for l, n in enumerate(nets):
        o = l * per
        n.infer_sequence_dev(fr[o:].data_ptr(), per + 1, 720, 1280, 1.0, 12, False, 0, rgb[o:].data_ptr(), mx[o:].data_ptr())
    for n in nets: n.sync()
step(); step()
t0 = time.perf_counter()
for _ in range(3): step()
dt = (time.perf_counter() - t0) / 3
print("pairs %d lanes %d: %.1f pairs/s  %.2f ms/step" % (pairs, L, pairs / dt, dt * 1e3))
