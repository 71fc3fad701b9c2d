#!/usr/bin/env python3
"""Does a re-planned arena run slower?  depth_anything on the bench's 32 x 1080p batch (frames resident in HBM): fresh context; after a one-frame 720p
call on the same context (arena re-planned twice); on a second fresh context created while the first still exists: python tools/replan_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prisma_amd import engine, synth  # noqa: E402

B, H, W = 32, 1080, 1920
cfg = synth.DEPTH_CFGS["vitl"]
w = synth.cached_weights("depth", cfg, 1234)
d_frames = torch.from_numpy(synth.frame_pair_sequence(B, H, W, seed=1000)).cuda()
d_rgb = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda")
scal = torch.zeros((2, B), dtype=torch.float32, device="cuda")
f1 = torch.from_numpy(synth.frames(1, 720, 1280, seed=7)).cuda()
r1 = torch.empty((1, 720, 1280, 3), dtype=torch.uint8, device="cuda")


def big(net, n=6):
    for _ in range(2):
        net.infer_dev(d_frames.data_ptr(), B, H, W, 0, d_rgb.data_ptr(), scal[0].data_ptr(), scal[1].data_ptr(), True); net.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        net.infer_dev(d_frames.data_ptr(), B, H, W, 0, d_rgb.data_ptr(), scal[0].data_ptr(), scal[1].data_ptr(), True); net.sync()
    return (time.perf_counter() - t0) / n * 1e3


def small(net):
    for _ in range(3):
        net.infer_dev(f1.data_ptr(), 1, 720, 1280, 0, r1.data_ptr(), scal[0].data_ptr(), scal[1].data_ptr(), True); net.sync()


a = engine.DepthAnything(w, cfg, device=0, max_batch=B, precision=1)
print(f"fresh context, 32 x 1080p:                          {big(a):7.2f} ms", flush=True)
print(f"again:                                              {big(a):7.2f} ms", flush=True)
small(a)
print(f"after a one-frame 720p call (arena re-planned):     {big(a):7.2f} ms", flush=True)
print(f"again:                                              {big(a):7.2f} ms", flush=True)
b = engine.DepthAnything(w, cfg, device=0, max_batch=B, precision=1)
print(f"a second fresh context beside the first:            {big(b):7.2f} ms", flush=True)
print(f"the first one again:                                {big(a):7.2f} ms", flush=True)
a.close()
c = engine.DepthAnything(w, cfg, device=0, max_batch=B, precision=1)
print(f"a third fresh context after closing the first:      {big(c):7.2f} ms", flush=True)
