#!/usr/bin/env python3
"""PCIe-inclusive rate of the bench clip (32 x 1080p, page-locked frames in / results out) by entry point: blocking calls (one host thread per band
or one band after the other) against streamed submissions (pb_*_submit_* / pb_wait, two clips in flight per band), per band and for both bands:
python tools/pcie_chunk_bench.py"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prisma_amd import engine, synth  # noqa: E402

B, H, W = 32, 1080, 1920
cfg = synth.DEPTH_CFGS["vitl"]
dn = engine.DepthAnything(synth.cached_weights("depth", cfg, 1234), cfg, device=0, max_batch=B, precision=1)
fn = engine.FlowRaft(synth.cached_weights("raft", 4321), device=0, precision=1)
frames = synth.frame_pair_sequence(B, H, W, seed=1000)
sh, sw = engine.flow_out_size(H, W, 0.75)
pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory()      # noqa: E731
hf = torch.from_numpy(frames).pin_memory()
sets = [dict(rgb=pin((B, H, W, 3), torch.uint8), mn=pin((B,), torch.float32), mx=pin((B,), torch.float32), frgb=pin((B - 1, 1, sh, sw, 3), torch.uint8),
             fmx=pin((B - 1, 1), torch.float32)) for _ in range(2)]
N = 6


def d_block(k):
    dn.infer_batch(hf.numpy(), want_depth=False, want_rgb=True, flip=True, out_rgb=sets[k % 2]["rgb"].numpy())


def f_block(k):
    fn.infer_sequence(hf.numpy(), scale=0.75, iters=12, backward=False, want_flow=False, want_rgb=True, out_rgb=sets[k % 2]["frgb"].numpy())


def d_sub(k):
    s = sets[k % 2]
    dn.submit_batch(hf.numpy(), out_rgb=s["rgb"].numpy(), out_min=s["mn"].numpy(), out_max=s["mx"].numpy())


def f_sub(k):
    s = sets[k % 2]
    fn.submit_sequence(hf.numpy(), scale=0.75, iters=12, out_rgb=s["frgb"].numpy(), out_max=s["fmx"].numpy())


def timed(name, body, frames_per_clip=B):
    body(2)
    t0 = time.perf_counter()
    body(N)
    dt = (time.perf_counter() - t0) / N
    print(f"{name:72s} {dt * 1e3:7.1f} ms per clip  {frames_per_clip / dt:6.1f} frames/s", flush=True)


def blocking(bands):
    def body(n):
        for k in range(n):
            ths = [threading.Thread(target=b, args=(k,)) for b in bands[1:]]
            for t in ths:
                t.start()
            bands[0](k)
            for t in ths:
                t.join()
    return body


def streamed(subs, nets, depth=2):
    def body(n):
        for k in range(n):
            for s in subs:
                s(k)
            if k >= depth - 1:
                for net in nets:
                    net.wait()
        for _ in range(min(depth - 1, n)):
            for net in nets:
                net.wait()
    return body


for dc, fc in ((0, 0), (16, 16)):
    dn.set_option("host_chunk", dc); fn.set_option("host_chunk", fc)
    print(f"# depth chunk {dc or 'default (max_batch frames)'}, flow chunk {fc or 'default (32 pairs)'}", flush=True)
    timed("depth alone, blocking", blocking([d_block]))
    timed("depth alone, streamed (2 in flight)", streamed([d_sub], [dn]))
    timed("flow alone, blocking", blocking([f_block]))
    timed("flow alone, streamed (2 in flight)", streamed([f_sub], [fn]))
    timed("both bands, blocking, one host thread per band", blocking([d_block, f_block]))
    timed("both bands, streamed from one thread (2 in flight)", streamed([d_sub, f_sub], [dn, fn]))
    timed("both bands, submit + wait each clip (1 in flight)", streamed([d_sub, f_sub], [dn, fn], depth=1))
