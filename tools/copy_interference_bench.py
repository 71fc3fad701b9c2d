#!/usr/bin/env python3
"""Do PCIe copies slow the bands' kernels down?  The bench step (both bands at once, frames resident in HBM) alone, then with INDEPENDENT page-locked
H2D / D2H traffic of the host path's volume (2 x 199 MB up, 199 + 109 MB down per step) running beside it on two other streams with no dependency on
the bands: python tools/copy_interference_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from prisma_amd import engine, synth  # noqa: E402
from prisma_amd.power import PowerSampler  # noqa: E402

B, H, W = 32, 1080, 1920
cfg = synth.DEPTH_CFGS["vitl"]
dn = engine.DepthAnything(synth.cached_weights("depth", cfg, 1234), cfg, device=0, max_batch=B, precision=1)
fn = engine.FlowRaft(synth.cached_weights("raft", 4321), device=0, precision=1)
frames = synth.frame_pair_sequence(B, H, W, seed=1000)
d_frames = torch.from_numpy(frames).cuda()
sh, sw = engine.flow_out_size(H, W, 0.75)
d_rgb = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda")
f_rgb = torch.empty((B - 1, sh, sw, 3), dtype=torch.uint8, device="cuda")
scal = torch.zeros((3, B), dtype=torch.float32, device="cuda")
hf = torch.from_numpy(frames).pin_memory()
h_out = torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory()
h_out2 = torch.empty((B - 1, sh, sw, 3), dtype=torch.uint8).pin_memory()
scratch_in = [torch.empty_like(d_frames) for _ in range(2)]
scratch_out = torch.empty_like(d_rgb)
s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()


def step(copies):
    if copies:
        with torch.cuda.stream(s_in):
            scratch_in[0].copy_(hf, non_blocking=True)
            scratch_in[1].copy_(hf, non_blocking=True)
        with torch.cuda.stream(s_out):
            h_out.copy_(scratch_out, non_blocking=True)
            h_out2.copy_(f_rgb, non_blocking=True)
    engine.run_concurrently([
        (dn, lambda: dn.infer_dev(d_frames.data_ptr(), B, H, W, 0, d_rgb.data_ptr(), scal[0].data_ptr(), scal[1].data_ptr(), True)),
        (fn, lambda: fn.infer_sequence_dev(d_frames.data_ptr(), B, H, W, 0.75, 12, False, 0, f_rgb.data_ptr(), scal[2].data_ptr()))])
    if copies:
        s_in.synchronize(); s_out.synchronize()


for copies in (False, True, False, True):
    step(copies); step(copies)
    torch.cuda.synchronize()
    with PowerSampler() as ps:
        t0 = time.perf_counter()
        for _ in range(6):
            step(copies)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    w = ps.window(t0 + 0.1 * (t1 - t0), t1)
    print(f"{'with independent PCIe copies beside' if copies else 'bands alone                        '}: {(t1 - t0) / 6 * 1e3:7.1f} ms per step, {w['avg_power_w']} W, {w['avg_sclk_mhz']} MHz", flush=True)
# the copies alone
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(6):
    with torch.cuda.stream(s_in):
        scratch_in[0].copy_(hf, non_blocking=True); scratch_in[1].copy_(hf, non_blocking=True)
    with torch.cuda.stream(s_out):
        h_out.copy_(scratch_out, non_blocking=True); h_out2.copy_(f_rgb, non_blocking=True)
    s_in.synchronize(); s_out.synchronize()
print(f"the copies alone: {(time.perf_counter() - t0) / 6 * 1e3:7.1f} ms per step ({2 * hf.numel() / 1e6:.0f} MB up, {(h_out.numel() + h_out2.numel()) / 1e6:.0f} MB down)")
