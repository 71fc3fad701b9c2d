"""flow_raft leg under a forced conv tile: python tools/raft_tile_bench.py <conv_tile code> [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prisma_amd import engine, synth
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
net = engine.FlowRaft(synth.raft_weights(seed=4321))
net.set_option("conv_tile", tile)
fr = torch.from_numpy(synth.frame_pair_sequence(pairs + 1, 720, 1280, seed=50)).cuda()
rgb = torch.empty((pairs, 720, 1280, 3), dtype=torch.uint8, device="cuda"); mx = torch.empty((pairs,), dtype=torch.float32, device="cuda")
def step():
    net.infer_sequence_dev(fr.data_ptr(), pairs + 1, 720, 1280, 1.0, 12, False, 0, rgb.data_ptr(), mx.data_ptr()); net.sync()
step(); net.set_profiling(True)
t0 = time.perf_counter()
for _ in range(3): step()
dt = (time.perf_counter() - t0) / 3
print("conv_tile %d: %.1f pairs/s  %.2f ms/step  %s" % (tile, pairs / dt, dt * 1e3, {s["name"]: round(s["ms"], 2) for s in net.kernel_stats()}))
