#!/usr/bin/env python3
"""Per-step wall times of the bench step (both bands at once, frames resident in HBM) from a cold start, with the socket power and clock per step:
is there a ramp, a drift or a periodic throttle behind the box-to-box spread?  python tools/step_trace.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prisma_amd import engine, synth  # noqa: E402
from prisma_amd.power import PowerSampler  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B, H, W = 32, 1080, 1920
cfg = synth.DEPTH_CFGS["vitl"]
dn = engine.DepthAnything(synth.cached_weights("depth", cfg, 1234), cfg, device=0, max_batch=B, precision=1)
fn = engine.FlowRaft(synth.cached_weights("raft", 4321), device=0, precision=1)
d_frames = torch.from_numpy(synth.frame_pair_sequence(B, H, W, seed=1000)).cuda()
sh, sw = engine.flow_out_size(H, W, 0.75)
d_rgb = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda")
f_rgb = torch.empty((B - 1, sh, sw, 3), dtype=torch.uint8, device="cuda")
scal = torch.zeros((3, B), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
marks = []
with PowerSampler() as ps:
    for i in range(N):
        t0 = time.perf_counter()
        engine.run_concurrently([
            (dn, lambda: dn.infer_dev(d_frames.data_ptr(), B, H, W, 0, d_rgb.data_ptr(), scal[0].data_ptr(), scal[1].data_ptr(), True)),
            (fn, lambda: fn.infer_sequence_dev(d_frames.data_ptr(), B, H, W, 0.75, 12, False, 0, f_rgb.data_ptr(), scal[2].data_ptr()))])
        marks.append((t0, time.perf_counter()))
        if i == N // 2:
            time.sleep(2.0)            # an idle gap in the middle: does the next step start slow?
for i, (a, b) in enumerate(marks):
    w = ps.window(a, b)
    print(f"step {i:3d}: {(b - a) * 1e3:7.2f} ms  {w['avg_power_w']} W  {w['avg_sclk_mhz']} MHz")
