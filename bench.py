#!/usr/bin/env python3
"""bench.py - frames/sec of the depth_anything band (ViT-L/14 + DPT) on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by torch.distributed.run, one rank per GPU.  A step = one pass of the hot path
  (uint8 1080p frames resident in HBM -> depth_anything: heat-encoded uint8 frames + per-frame min/max;
  flow_raft: HSV-encoded flow of the consecutive pairs + max displacement; both in HBM, the two bands
  enqueued at once on their own streams) over one clip of synthetic frames per GPU.  Frames shard by rank with no data-path
  collective; the only exchange is the all-gather of the per-frame min/max scalars (RCCL).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL between processes needs dmabuf IPC on this driver (the image exports it; kept here for a shell that does not)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402  (plumbing: device sync + torch.distributed over RCCL)

GFLOP_PER_FRAME = 2583.1          # ViT-L @ 518x924, multiply-add = 2 FLOP
PEAK_F16_TFLOPS = 2500.0          # MI355X dense fp16/bf16 MFMA (MI355X_MICROARCH.md)


class Ranks:
    """torch.distributed glue of the bench: one process per GPU over RCCL ("nccl"); PRISMA_BENCH_BACKEND=gloo runs the same
    control flow with CPU-side collectives so that world_size > 1 can be exercised on a box with fewer GPUs than ranks."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.gpu = torch.cuda.is_available()              # False only in the CPU unit test of this class
        ngpu = max(torch.cuda.device_count(), 1) if self.gpu else 1
        self.device = int(os.environ.get("LOCAL_RANK", "0")) % ngpu
        self.backend = os.environ.get("PRISMA_BENCH_BACKEND", "nccl")
        self.dist = None
        if self.gpu:
            torch.cuda.set_device(self.device)
        # PRISMA_FORCE_DIST=1: build the process group even for one rank, so the RCCL calls of the N > 1 path (init, barrier,
        # all-reduce, all-gather) execute on a single-GPU box (tests/test_gpu_edges.py)
        if self.world > 1 or os.environ.get("PRISMA_FORCE_DIST", "") == "1":
            import torch.distributed as dist
            self.dist = dist
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.device))
            else:
                dist.init_process_group(self.backend)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        if self.gpu:
            torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_gather(self, out, src):
        """out [world, ...] <- src [...] of every rank (device tensors; staged through the host for a CPU backend)"""
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(out.view(-1), src.view(-1))
        else:
            o = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(o.view(-1), src.detach().cpu().view(-1))
            out.copy_(o)

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def cpu_baseline(weights, cfg, rweights, frames, flow_scale, flow_iters):
    """Oracle (CPU restatement of the reference, torch fp32 on the host cores) on a bounded sample: one frame through the depth
    band and one frame pair through the flow band (as the reference's infer does it: forward and backward in one batch of 2)."""
    from oracle import depth_oracle as O
    from oracle import raft_oracle as RO
    cores = min(os.cpu_count() or 1, 32)      # torch CPU GEMMs stop scaling (and regress) past ~32 threads
    torch.set_num_threads(cores)
    H, W = frames.shape[1:3]
    t0 = time.time()
    d = O.infer(weights, frames[0], cfg.depth, cfg.heads)
    O.encode_depth_video(d, flip=True)
    t1 = time.time()
    RO.infer_pair(rweights, frames[0], frames[1], scale=flow_scale, iters=flow_iters, backward=False)
    t2 = time.time()
    return {"value": round(1.0 / (t2 - t0), 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 frame {W}x{H} through oracle/depth_oracle.py ({t1 - t0:.1f} s) + 1 frame pair through oracle/raft_oracle.py "
                      f"(--scale {flow_scale}, {flow_iters} iterations, FORWARD direction only, like the GPU leg: {t2 - t1:.1f} s); "
                      f"torch fp32 CPU restatements of the reference, {cores} threads"}


def flow_leg(args, R):
    local_rank, world, rank = R.device, R.world, R.rank
    """Secondary line: flow_raft on 1280x720 frames, 12 GRU iterations, 8 forward pairs per GPU per step
    (BASELINE.json configs[2]); no --scale so the reference's 1559.6 GFLOP/pair-direction figure applies."""
    from prisma_amd import engine, synth
    H, W, pairs, iters = 720, 1280, args.flow_pairs, 12
    wts = synth.raft_weights(seed=4321)
    net = engine.FlowRaft(wts, device=local_rank, precision=args.precision)
    frames = synth.frame_pair_sequence(pairs + 1, H, W, seed=50 + rank)
    d_frames = torch.from_numpy(frames).cuda()
    d_rgb = torch.empty((pairs, H, W, 3), dtype=torch.uint8, device="cuda")
    d_mx = torch.empty((pairs,), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    def step():
        net.infer_sequence_dev(d_frames.data_ptr(), pairs + 1, H, W, 1.0, iters, False, 0, d_rgb.data_ptr(), d_mx.data_ptr())
        net.sync()

    step()
    R.barrier()
    steps = max(1, args.steps // 2)
    t0 = time.perf_counter()
    for _ in range(steps):                      # timed without per-kernel events: ~300 small launches per step
        step()
    R.barrier()
    dt = R.max_over_ranks(time.perf_counter() - t0)
    net.set_profiling(timing=True)              # the per-kernel breakdown comes from extra, untimed steps
    fam = {}
    for _ in range(steps):
        step()
        for s in net.kernel_stats():
            f = fam.setdefault(s["name"], dict(ms=0.0, flops=0.0))
            f["ms"] += s["ms"]; f["flops"] += s["flops"]
    mx = d_mx.cpu().numpy()
    assert np.isfinite(mx).all() and (mx > 0).all()
    out = {"metric": "frame-pairs/sec (flow_raft, 1280x720, 12 iters, forward)", "value": round(world * pairs * steps / dt, 3),
           "unit": "pairs/s", "ms_per_step": round(dt / steps * 1e3, 3), "pairs_per_step_per_gpu": pairs,
           "model_tflops": round(pairs * steps / dt * 1559.6 / 1e3, 2),
           "kernel_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in sorted(fam.items())},
           "kernel_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in fam.items() if v["flops"] > 0 and v["ms"] > 0}}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import raft_oracle as R
        cores = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(cores)
        t0 = time.time()
        R.infer_pair(wts, frames[0], frames[1], scale=1.0, iters=iters, backward=False)
        dt = time.time() - t0
        out["cpu_baseline"] = {"value": round(1.0 / dt, 4), "unit": "pairs/s", "cores": cores, "kind": "port",
                               "sample": f"1 pair 1280x720, forward direction only (like the GPU leg), oracle/raft_oracle.py, {dt:.1f} s wall"}
    net.close()
    return out


def gmflow_leg(args, R):
    """flow_gmflow (prisma's default flow band) on the headline flow shape: the forward pairs of a 1080p clip at the band's
    default --scale 0.75 (816 x 1440 after padding to /16: 18360 tokens per frame, 51 x 90 attention windows), args.gmflow_pairs per step."""
    local_rank, world, rank = R.device, R.world, R.rank
    from prisma_amd import engine, synth
    H, W, pairs = 1080, 1920, args.gmflow_pairs
    net = engine.FlowGMFlow(synth.gmflow_weights(seed=2468), device=local_rank, precision=args.precision)
    frames = synth.frame_pair_sequence(pairs + 1, H, W, seed=150 + rank)
    d_frames = torch.from_numpy(frames).cuda()
    sh, sw = engine.flow_out_size(H, W, 0.75)
    d_rgb = torch.empty((pairs, sh, sw, 3), dtype=torch.uint8, device="cuda")
    d_mx = torch.empty((pairs,), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    def step():
        net.infer_sequence_dev(d_frames.data_ptr(), pairs + 1, H, W, 0.75, 1, False, 0, d_rgb.data_ptr(), d_mx.data_ptr())
        net.sync()

    step()
    R.barrier()
    steps = max(1, args.steps // 2)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    R.barrier()
    dt = R.max_over_ranks(time.perf_counter() - t0)
    net.set_profiling(timing=True)
    fam = {}
    step()
    for s in net.kernel_stats():
        f = fam.setdefault(s["name"], dict(ms=0.0, flops=0.0, exec=0.0))
        f["ms"] += s["ms"]; f["flops"] += s["flops"]; f["exec"] += s["exec_flops"]
    mx = d_mx.cpu().numpy()
    assert np.isfinite(mx).all() and (mx > 0).all()
    net.close()
    return {"metric": "frame-pairs/sec (flow_gmflow, 1080p x 0.75, forward)", "value": round(world * pairs * steps / dt, 3), "unit": "pairs/s",
            "ms_per_step": round(dt / steps * 1e3, 3), "pairs_per_step_per_gpu": pairs, "precision": PREC_NAME[args.precision],
            "gflop_per_pair": round(sum(v["flops"] for v in fam.values()) / pairs / 1e9, 1),
            "kernel_ms_per_step": {k: round(v["ms"], 3) for k, v in sorted(fam.items())},
            "kernel_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in fam.items() if v["flops"] > 0 and v["ms"] > 0},
            "kernel_executed_tflops": {k: round(v["exec"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in fam.items() if v["exec"] > 0 and v["ms"] > 0}}


def mask_leg(args, R):
    local_rank, world, rank = R.device, R.world, R.rank
    """Third line: the mask band (SOLOv2 R-101 FPN) on 1920x1080 frames, args.mask_frames per GPU per step, resident in
    HBM (BASELINE.json configs[4] runs it beside depth and flow).  FLOP count: the convolution / GEMM launches'
    own multiply-adds (the dynamic convolution depends on how many grid cells fire)."""
    from prisma_amd import engine, synth
    H, W, B = 1080, 1920, args.mask_frames
    cfg = synth.MASK_CFGS["r101"]
    wts = synth.solov2_weights(cfg)
    net = engine.MaskMMDet(wts, cfg, device=local_rank, max_batch=min(B, 32), precision=args.precision)      # chunks of 32: 580 fps against 484 at 8
    frames = synth.frames(B, H, W, seed=70 + rank)
    d_frames = torch.from_numpy(frames).cuda()
    d_out = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda")
    keep = [synth.COCO_CLASSES.index(c) for c in synth.BAND_CLASSES]
    torch.cuda.synchronize()

    def step():
        net.infer_batch_dev(d_frames.data_ptr(), B, H, W, 0.5, keep, d_out.data_ptr())
        net.sync()

    step()
    R.barrier()
    steps = max(1, args.steps // 2)
    t0 = time.perf_counter()
    for _ in range(steps):                      # timed without per-kernel events: ~300 small launches per step
        step()
    R.barrier()
    dt = R.max_over_ranks(time.perf_counter() - t0)
    net.set_profiling(timing=True)              # the per-kernel breakdown comes from extra, untimed steps
    fam = {}
    for _ in range(steps):
        step()
        for s in net.kernel_stats():
            f = fam.setdefault(s["name"], dict(ms=0.0, flops=0.0))
            f["ms"] += s["ms"]; f["flops"] += s["flops"]
    drawn = int((d_out[:, :, :, 0] != 0).any(dim=2).any(dim=1).sum().item())
    inst = [len(net.instances(b)[0]) for b in range(B)]
    cand = [net.instances(b)[3] for b in range(B)]
    out = {"metric": "frames/sec (mask_mmdet SOLOv2 R-101, 1080p)", "value": round(world * B * steps / dt, 3), "unit": "frames/s",
           "ms_per_step": round(dt / steps * 1e3, 3), "frames_per_step_per_gpu": B,
           "gflop_per_frame": round(sum(v["flops"] for v in fam.values()) / (steps * B) / 1e9, 1),
           "frames_with_masks": drawn, "mean_candidates": round(float(np.mean(cand)), 1), "mean_instances": round(float(np.mean(inst)), 1),
           "kernel_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in sorted(fam.items())},
           "kernel_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in fam.items() if v["flops"] > 0 and v["ms"] > 0}}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import solov2_oracle as SO
        cores = min(os.cpu_count() or 1, 32)
        torch.set_num_threads(cores)
        t0 = time.time()
        SO.infer(wts, cfg, frames[0], synth.COCO_CLASSES, synth.BAND_CLASSES, 0.5)
        dt = time.time() - t0
        out["cpu_baseline"] = {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
                               "sample": f"1 frame 1920x1080, oracle/solov2_oracle.py, {dt:.1f} s wall"}
    net.close()
    return out


def pipeline_leg(args, R):
    local_rank, world, rank = R.device, R.world, R.rank
    """BASELINE.json configs[4]: every frame of a 1080p clip goes through depth_anything, flow_raft (forward pairs, the band's
    default --scale 0.75, 12 iterations) and mask_mmdet, fused pre / post-processing, frames resident in HBM.  One step =
    args.pipeline_frames frames through the three bands back to back on this rank's GPU."""
    from prisma_amd import engine, synth
    H, W, B = 1080, 1920, args.pipeline_frames
    dn = engine.DepthAnything(synth.depth_anything_weights("vitl", seed=1234), "vitl", device=local_rank, max_batch=B, precision=args.precision)
    fn = engine.FlowRaft(synth.raft_weights(seed=4321), device=local_rank, precision=args.precision)
    mcfg = synth.MASK_CFGS["r101"]
    mn = engine.MaskMMDet(synth.solov2_weights(mcfg), mcfg, device=local_rank, max_batch=min(B, 32), precision=args.precision)
    frames = torch.from_numpy(synth.frame_pair_sequence(B, H, W, seed=90 + rank)).cuda()
    d_rgb = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda")
    d_mm = torch.empty((2, B), dtype=torch.float32, device="cuda")
    sh, sw = engine.flow_out_size(H, W, 0.75)
    f_rgb = torch.empty((B - 1, sh, sw, 3), dtype=torch.uint8, device="cuda")
    f_mx = torch.empty((B - 1,), dtype=torch.float32, device="cuda")
    m_out = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda")
    keep = [synth.COCO_CLASSES.index(c) for c in synth.BAND_CLASSES]

    def step():             # the bands are independent: all three are enqueued on their own streams and share the GPU
        dn.infer_dev(frames.data_ptr(), B, H, W, 0, d_rgb.data_ptr(), d_mm[0].data_ptr(), d_mm[1].data_ptr(), True)
        fn.infer_sequence_dev(frames.data_ptr(), B, H, W, 0.75, 12, False, 0, f_rgb.data_ptr(), f_mx.data_ptr())
        mn.infer_batch_dev(frames.data_ptr(), B, H, W, 0.5, keep, m_out.data_ptr())
        dn.sync(); fn.sync(); mn.sync()

    def timed(fn_step, steps):
        fn_step()
        R.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn_step()
        R.barrier()
        return R.max_over_ranks(time.perf_counter() - t0)

    steps = max(1, args.steps // 2)
    dt = timed(step, steps)
    # PCIe-inclusive figure of configs[4] (one-GPU runs): the same clip through the three HOST-pointer entry points - page-locked frames in,
    # page-locked results out, the library's default chunking (depth max_batch frames, flow 32 pairs + a halo frame, mask chunks of max_batch
    # frames), one host thread per band since the calls block until their results are in host memory - compared byte for byte with the
    # HBM-resident leg's results
    host_fps = None
    if world == 1 and args.host_clips > 0:
        import threading
        hf = frames.cpu().pin_memory()
        h_d = torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory()
        h_f = torch.empty((B - 1, 1, sh, sw, 3), dtype=torch.uint8).pin_memory()
        h_m = torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory()

        def clip():
            ths = [threading.Thread(target=lambda: dn.infer_batch(hf.numpy(), want_depth=False, want_rgb=True, flip=True, out_rgb=h_d.numpy())),
                   threading.Thread(target=lambda: fn.infer_sequence(hf.numpy(), scale=0.75, iters=12, backward=False, want_flow=False, want_rgb=True, out_rgb=h_f.numpy())),
                   threading.Thread(target=lambda: mn.infer_batch(hf.numpy(), 0.5, keep, out=h_m.numpy()))]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        clip()
        assert bool((h_d == d_rgb.cpu()).all()) and bool((h_f[:, 0] == f_rgb.cpu()).all()) and bool((h_m == m_out.cpu()).all()), \
            "host-pointer results of the three-band pipeline differ from the HBM-resident leg's"
        t1 = time.perf_counter()
        for _ in range(args.host_clips):
            clip()
        host_fps = B * args.host_clips / (time.perf_counter() - t1)
    for n_ in (dn, fn, mn):
        n_.close()
    return {"metric": "frames/sec (depth_anything + flow_raft + mask_mmdet on every 1080p frame)", "value": round(world * B * steps / dt, 3),
            "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3), "frames_per_step_per_gpu": B,
            "pcie_inclusive_fps": round(host_fps, 2) if host_fps else None,
            "note": "the three bands enqueued together on their own streams (tools/pipeline_order_bench.py: 121 frames/s against 116 one after the other); flow at --scale 0.75 (816 x 1440), forward pairs only; "
                    "pcie_inclusive_fps = the same clip through the three host-pointer entry points from / into page-locked memory, one host thread per band, results byte-identical"}


def pmc_traffic(symbol):
    """HBM bytes per launch of `symbol` from the committed rocprofv3 PMC summary of THIS command (tools/pmc_summary.py; separate
    FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md).  PMC passes cannot run inside the
    timed bench, so the figure is read from profiles/ (newest round first)."""
    import glob
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    want = symbol.replace(" ", "")
    for path in sorted(glob.glob(os.path.join(here, "r??[a-z]_pmc_traffic.json")), reverse=True):      # newest round / letter first
        for kname, v in json.load(open(path))["kernels"].items():
            if want in kname.replace(" ", ""):
                return round(v["fetch_bytes"] + v["write_bytes"]), "profiles/" + os.path.basename(path)
    return None, None


def pmc_clocks(fam_keys):
    """per-symbol effective shader clock (GHz) of the bench's kernel families from the newest committed GRBM_GUI_ACTIVE pass over
    `bench.py --sequential-only` (tools/pmc_clock.py; counter passes cannot run inside the timed bench): {family: GHz}, source file"""
    import glob
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for path in sorted(glob.glob(os.path.join(here, "r??[a-z]_clock_per_symbol.json")), reverse=True):
        ks = json.load(open(path))["kernels"]
        out = {}
        for fk in fam_keys:
            sym = fk.split("/", 1)[1]
            want = SYMBOLS.get(sym, sym).replace(" ", "").rstrip(">")      # (the attention symbol carries one more template argument than the family name)
            hit = [v for k, v in ks.items() if want in k.replace(" ", "")]
            if hit and "(" != want[:1]:
                out[fk] = round(sum(h["effective_clock_ghz"] * h["total_ms"] for h in hit) / sum(h["total_ms"] for h in hit), 3)
        return out, "profiles/" + os.path.basename(path)
    return None, None


# GEMM-shaped launches are keyed by the kernel symbol itself (pb_kernel_stat.name = what rocprofv3 prints); the other families:
SYMBOLS = {"attention": "attnq_kernel<1, 2, 0, false, 8>", "layernorm": "layernorm_kernel", "elementwise": "(bilinear, instance-norm, lookup, ... kernels)",
           "prepost": "(pre- / post-processing kernels)"}
PREC_NAME = {0: "f16", 1: "split-f16"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="frames of the clip per GPU per step")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--encoder", default="vitl")
    ap.add_argument("--flow-scale", type=float, default=0.75, help="flow_raft --scale (the band's default)")
    ap.add_argument("--flow-iters", type=int, default=12, help="GRU iterations (BASELINE.json configs[2])")
    ap.add_argument("--precision", type=int, default=1, choices=(0, 1),
                    help="pb_precision of the timed `value`: 1 (default) = split-fp16, the mode that meets north_star's 1e-3 tolerance on every "
                         "reference vector (what the parity tests assert and the band scripts use); 0 = one fp16 MFMA pass per GEMM (faster, "
                         "1.3-2.2e-3 max-norm error).  The other mode is timed too (child process) and reported under `other_precision`")
    ap.add_argument("--one-precision", action="store_true", help="skip the second precision mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="(internal) run the CPU baseline of this configuration and print it as one JSON object: the default run starts this as a "
                         "child process next to its GPU work when the host has cores to spare")
    ap.add_argument("--gemm-tile", type=int, default=0, help="A/B switch: 0 auto, 1 128x128, 2 ping-pong 256x256, 9 256x64")
    ap.add_argument("--conv-tile", type=int, default=0)
    ap.add_argument("--all-legs", action="store_true", help="also run the per-band legs below with their usual sizes (flow 720p, mask, pipeline, PCIe)")
    ap.add_argument("--latency", action="store_true", default=True,
                    help="also time one 1280x720 frame at batch 1 (BASELINE configs[1]; on by default since round 4: eight batch-1 calls, ~0.1 s)")
    ap.add_argument("--no-latency", dest="latency", action="store_false")
    ap.add_argument("--sequential-only", action="store_true",
                    help="time the step the way rounds 1-5 did: depth band, then flow band, every launch bracketed by HIP events.  The default timed region "
                         "enqueues both bands of a step at once on their two ctx streams (prisma_amd.engine.run_concurrently) without per-launch events and "
                         "takes the per-symbol attribution (`sequential`, `roofline`, `kernel_ms_per_step`) from a second, labelled pass of this kind; "
                         "`rocprofv3 --kernel-trace --stats -- python bench.py --sequential-only` reproduces that pass's per-symbol averages")
    ap.add_argument("--seq-steps", type=int, default=5, help="steps of the labelled sequential pass behind `sequential` / `roofline` (default timed region only)")
    ap.add_argument("--no-clock", action="store_true", help="skip the effective-clock probe behind roofline.effective_clock_ghz")
    ap.add_argument("--host-clips", type=int, default=6,
                    help="clips pushed through the host-pointer entry points of both bands (page-locked frames in, page-locked results out) for "
                         "`pcie_inclusive_fps`; 0 = skip")
    ap.add_argument("--pipeline-frames", type=int, default=0, help="frames per step of the three-band pipeline leg")
    ap.add_argument("--mask-frames", type=int, default=0, help="frames per step of the mask_mmdet leg")
    ap.add_argument("--gmflow-pairs", type=int, default=0, help="frame pairs per GPU per step of the flow_gmflow leg (1080p x 0.75)")
    ap.add_argument("--flow-pairs", type=int, default=0, help="frame pairs per GPU per step of the 720p flow_raft leg (BASELINE configs[2])")
    args = ap.parse_args()
    if args.all_legs:
        args.latency = True
        args.pipeline_frames, args.mask_frames, args.flow_pairs = 32, 32, 8
        args.gmflow_pairs = args.gmflow_pairs or 15

    if args.cpu_baseline_only:
        from prisma_amd import synth
        cfg = synth.DEPTH_CFGS[args.encoder]
        fr = synth.frame_pair_sequence(args.batch, args.height, args.width, seed=1000)[:2]     # the parent's clip, its first two frames
        print(json.dumps(cpu_baseline(synth.cached_weights("depth", cfg, 1234), cfg, synth.cached_weights("raft", 4321), fr, args.flow_scale,
                                      args.flow_iters)))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the bands engine has no CPU path")
    R = Ranks()
    rank, local_rank, world = R.rank, R.device, R.world

    from prisma_amd import engine, synth
    cfg = synth.DEPTH_CFGS[args.encoder]
    if world > 1:
        # one rank generates the seeded weights (~10 s of one core for ViT-L), the others map them from the node's tmpfs (synth.cached_weights):
        # an 8-rank launch does not spend its start-up with eight processes competing for the host cores over identical tensors
        if rank == 0:
            synth.cached_weights("depth", cfg, 1234)
            synth.cached_weights("raft", 4321)
        R.barrier()
        weights = synth.cached_weights("depth", cfg, 1234)
        rweights = synth.cached_weights("raft", 4321)
    else:
        # through the cache as well: the other-precision child process (and the CPU-baseline child) map them instead of spending ~10 s of one
        # core on the same tensors again
        weights = synth.cached_weights("depth", cfg, 1234)
        rweights = synth.cached_weights("raft", 4321)
    B, H, W = args.batch, args.height, args.width
    # one synthetic clip per rank (a seeded noise texture shifted by a known step per frame, so the flow is not degenerate),
    # resident in HBM before the timed region; one step = both bands over the whole clip
    frames = synth.frame_pair_sequence(B, H, W, seed=1000 + rank)
    d_frames = torch.from_numpy(frames).cuda()
    d_rgb = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda")
    sh, sw = engine.flow_out_size(H, W, args.flow_scale)
    f_rgb = torch.empty((B - 1, sh, sw, 3), dtype=torch.uint8, device="cuda")
    scal = torch.zeros((3, B), dtype=torch.float32, device="cuda")      # per-frame depth min, depth max, flow max displacement
    gathered = torch.empty((world, 3, B), dtype=torch.float32, device="cuda") if R.dist is not None else None
    torch.cuda.synchronize()

    def run_mode(prec, steps, warmup, extras):
        """K timed steps in one precision mode.  A step enqueues the depth band and the flow band at once, each on its ctx stream
        (engine.run_concurrently: the bands are independent and share the GPU), waits for both, then runs the 12-byte-per-frame scalar
        all-gather.  The per-symbol attribution comes from a labelled sequential pass after the timed region (bands one after the
        other, every launch bracketed by HIP events: a launch's duration is then the kernel's own, the same thing rocprofv3 reports);
        --sequential-only makes that pass the timed region, as in rounds 1-5.  Returns wall time, per-family launch records of the
        sequential pass, per-band completion times and the socket power over the timed region."""
        dn = engine.DepthAnything(weights, cfg, device=local_rank, max_batch=B, precision=prec)
        dn.set_option("gemm_tile", args.gemm_tile)
        dn.set_option("conv_tile", args.conv_tile)
        fn = engine.FlowRaft(rweights, device=local_rank, precision=prec)

        def depth_job():
            dn.infer_dev(d_frames.data_ptr(), B, H, W, 0, d_rgb.data_ptr(), scal[0].data_ptr(), scal[1].data_ptr(), True)

        def flow_job():
            fn.infer_sequence_dev(d_frames.data_ptr(), B, H, W, args.flow_scale, args.flow_iters, False, 0, f_rgb.data_ptr(), scal[2].data_ptr())

        band_s = [0.0, 0.0]

        def step_seq():                        # one band after the other: a launch's HIP-event duration is the kernel's own
            a = time.perf_counter()
            depth_job(); dn.sync()
            b = time.perf_counter()
            flow_job(); fn.sync()
            c = time.perf_counter()
            band_s[0] += b - a; band_s[1] += c - b
            if R.dist is not None:
                R.all_gather(gathered, scal)                              # the only exchange: 12 bytes per frame

        def step_ovl():                        # both bands of the step at once, each on its ctx stream (the engine's run_concurrently)
            done = engine.run_concurrently([(dn, depth_job), (fn, flow_job)])
            band_s[0] += done[0]; band_s[1] += done[1]
            if R.dist is not None:
                R.all_gather(gathered, scal)

        def timed(step, n):
            band_s[0] = band_s[1] = 0.0
            R.barrier()
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            R.barrier()
            t1 = time.perf_counter()
            return t0, t1

        def families():
            fam = {}
            for band, net_ in (("depth", dn), ("flow", fn)):
                for s in net_.kernel_stats():
                    fam[band + "/" + s["name"]] = {k: s[k] for k in ("ms", "flops", "exec_flops", "bytes", "launches")}
                net_.set_profiling(timing=False)
            return fam

        step = step_seq if args.sequential_only else step_ovl
        for _ in range(warmup):
            step()
        from prisma_amd.power import PowerSampler
        seq = None
        with PowerSampler(local_rank) as ps:
            if args.sequential_only:
                # every launch of the timed region is bracketed by HIP events on its band's stream; the records accumulate over the K
                # steps and are read once after the closing barrier, so no event query sits inside the timed region
                dn.set_profiling(timing=True, accumulate=True)
                fn.set_profiling(timing=True, accumulate=True)
            t0, t1 = timed(step, steps)
        dt = t1 - t0
        power = ps.window(t0, t1) if rank == 0 else None
        depth_s, flow_s = band_s[0], band_s[1]
        if args.sequential_only:
            fam = families()
            seq = {"dt": R.max_over_ranks(dt), "steps": steps, "depth_s": depth_s, "flow_s": flow_s}
        else:
            # the labelled sequential pass: not part of `value`; the per-symbol attribution of the same work on the same warm chip
            ns = max(1, min(args.seq_steps, steps))
            step_seq()
            dn.set_profiling(timing=True, accumulate=True)
            fn.set_profiling(timing=True, accumulate=True)
            s0, s1 = timed(step_seq, ns)
            fam = families()
            seq = {"dt": R.max_over_ranks(s1 - s0), "steps": ns, "depth_s": band_s[0], "flow_s": band_s[1]}
        dt = R.max_over_ranks(dt)
        sc = scal.cpu().numpy()
        assert np.isfinite(sc).all() and (sc[1] > sc[0]).all() and (sc[2, :B - 1] > 0).all(), "degenerate depth range / flow"
        res = {"dt": dt, "fam": fam, "depth_s": depth_s, "flow_s": flow_s, "seq": seq, "power": power}
        if extras and rank == 0:
            # PCIe-inclusive rate (never `value`; SURVEY 8(d) config 4: frames "resident in pinned host memory"): the SAME clip through the host-pointer
            # entry points of both bands (abi.hip: H2D, the band and D2H of a chunk on three streams) from page-locked frames into page-locked result
            # arrays, with the library's default chunking (a 32-frame clip is one chunk per band).
            if args.host_clips > 0 and world == 1:       # (one-GPU runs: with several ranks the host's PCIe / memory paths are shared and only rank 0 would be measuring)
                # Streamed clips (round 6): the asynchronous host-pointer entry points (pb_depth_submit_batch / pb_flow_submit_sequence / pb_wait) keep two
                # clips in flight per band - clip k + 1's uploads run under clip k's kernels, clip k's downloads under clip k + 1's - which is how a caller
                # that decodes a video feeds the engine; both bands are submitted from this one thread.  --sequential-only uses the blocking calls, one
                # band after the other.  Results are compared byte for byte with the HBM-resident leg's.
                pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory()      # noqa: E731
                hf = torch.from_numpy(frames).pin_memory()
                sets = [dict(rgb=pin((B, H, W, 3), torch.uint8), mn=pin((B,), torch.float32), mx=pin((B,), torch.float32),
                             frgb=pin((B - 1, 1, sh, sw, 3), torch.uint8), fmx=pin((B - 1, 1), torch.float32)) for _ in range(2)]

                def submit(k):
                    s = sets[k % 2]
                    dn.submit_batch(hf.numpy(), out_rgb=s["rgb"].numpy(), out_min=s["mn"].numpy(), out_max=s["mx"].numpy(), flip=True)
                    fn.submit_sequence(hf.numpy(), scale=args.flow_scale, iters=args.flow_iters, backward=False, out_rgb=s["frgb"].numpy(), out_max=s["fmx"].numpy())

                def blocking(k):
                    s = sets[k % 2]
                    dn.infer_batch(hf.numpy(), want_depth=False, want_rgb=True, flip=True, out_rgb=s["rgb"].numpy())
                    fn.infer_sequence(hf.numpy(), scale=args.flow_scale, iters=args.flow_iters, backward=False, want_flow=False, want_rgb=True, out_rgb=s["frgb"].numpy())

                def same(k):
                    s = sets[k % 2]
                    return bool((s["rgb"] == d_rgb.cpu()).all().item()) and bool((s["frgb"][:, 0] == f_rgb.cpu()).all().item())

                if args.sequential_only:
                    blocking(0)
                    assert same(0), "host-pointer results differ from the HBM-resident leg's"
                    t1 = time.perf_counter()
                    for k in range(args.host_clips):
                        blocking(k)
                    res["host_fps"] = world * B * args.host_clips / (time.perf_counter() - t1)
                else:
                    submit(0); dn.wait(); fn.wait()
                    assert same(0), "host-pointer results differ from the HBM-resident leg's"
                    n_clips = max(args.host_clips, 2)

                    def stream_clips(n):
                        submit(0)
                        for k in range(1, n):
                            submit(k)                  # clip k is enqueued before clip k - 1 is waited for
                            dn.wait(); fn.wait()
                        dn.wait(); fn.wait()
                    stream_clips(2)                    # untimed: the pipeline's steady state is what a video sees
                    t1 = time.perf_counter()
                    stream_clips(n_clips)
                    res["host_fps"] = world * B * n_clips / (time.perf_counter() - t1)
                    assert same(n_clips - 1) and np.isfinite(sets[(n_clips - 1) % 2]["mn"].numpy()).all() and (sets[(n_clips - 1) % 2]["fmx"].numpy() > 0).all()
                del hf, sets
            # latency of BASELINE.json configs[1]: one 1280x720 frame, batch 1 (outside the timed region above; AFTER the PCIe-inclusive leg: with this
            # leg in front of it the streamed clips ran 5 % slower on one box (275.7 against 262.4 ms per clip, profiles/r06n_*) - not the arena re-plan the
            # one-frame call causes (tools/replan_bench.py: 133.5 -> 133.7 ms), cause unknown, order chosen by measurement)
            if args.latency:
                # a context of its own with max_batch = 1 - what a one-frame-per-call caller creates (the band script with PRISMA_BATCH=1): such a
                # context lends its GEMM launches a split-K workspace (engine.h sk_ws_; 20-160 tiles for 256 CUs otherwise).  The 32-frame
                # context's figure for a one-frame call is reported beside it.
                f1 = torch.from_numpy(synth.frames(1, 720, 1280, seed=7)).cuda()
                r1 = torch.empty((1, 720, 1280, 3), dtype=torch.uint8, device="cuda")
                d1 = engine.DepthAnything(weights, cfg, device=local_rank, max_batch=1, precision=prec)
                for key, net_ in (("lat_b1", d1), ("lat_b1_big_ctx", dn)):
                    for i in range(13):
                        if i == 3:
                            torch.cuda.synchronize(); t1 = time.perf_counter()
                        net_.infer_dev(f1.data_ptr(), 1, 720, 1280, 0, r1.data_ptr(), scal[0].data_ptr(), scal[1].data_ptr(), True)
                        net_.sync()
                    res[key] = (time.perf_counter() - t1) / 10 * 1e3
                d1.close()
        dn.close(); fn.close()
        return res

    main_res = run_mode(args.precision, args.steps, args.warmup, True)
    # CPU baseline (rank 0 of a one-GPU run): the oracle on 32 host threads takes ~10 s and needs no GPU - on a host with cores to spare it
    # runs as a child process NEXT to the remaining GPU legs (clock probe, the other precision mode's child, optional legs) instead of after
    # them; the driver's wall clock around this command is then mostly GPU time.  It starts only now: beside the timed
    # region it cost the depth band 20 ms of launch gaps on one box (r05z, first attempt).  Smaller hosts run it at the end, as before.
    cpu_child = None
    if world == 1 and not args.no_cpu_baseline and (os.cpu_count() or 1) >= 48:
        import subprocess
        cpu_child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--batch", str(B), "--height", str(H), "--width", str(W),
                                      "--encoder", args.encoder, "--flow-scale", str(args.flow_scale), "--flow-iters", str(args.flow_iters)],
                                     stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    # effective shader clock under the band's dominant GEMM shape: per-tile s_memtime / s_memrealtime stamps of one
    # launch of the ping-pong kernel at the ViT's fc1 + GELU shape, right after the timed region (chip warm).  The 2.5 PF the fractions are
    # quoted against assume 2.4 GHz; on data the part is power-limited (profiles/r04a_clock_mfma_peak.txt: 1.63-1.65 GHz in a bare MFMA loop).
    clock_ghz = None
    if rank == 0 and not args.no_clock:
        try:
            import tempfile
            dump = os.path.join(tempfile.gettempdir(), "prisma_bench_gemm_dbg.%d.bin" % os.getpid())
            os.environ["PB_GEMM_DBG"] = dump
            ops = engine.Ops(local_rank)
            ops.gemm_bench(B * 2448, 4096, 1024, tile=2, epi=1, iters=3)
            ops.close()
            d = np.fromfile(dump, dtype=np.int64).reshape(-1, 8)
            d = d[d[:, 3] != 0]
            clock_ghz = float(np.median((d[:, 3] - d[:, 0]) / np.maximum((d[:, 5] - d[:, 4]) * 10.0, 1.0)))
            os.remove(dump)
        except Exception as e:      # noqa: BLE001 - a diagnostic: the line is complete without it
            print(f"[bench] clock probe skipped: {e}", file=sys.stderr)
        finally:
            os.environ.pop("PB_GEMM_DBG", None)
    # the other precision mode runs in a child process (single-GPU runs only): its launches then land in their own rocprofv3
    # per-process file, so the per-symbol averages of `rocprofv3 --stats -- python bench.py` stay those of ONE mode each
    other = None
    if not args.one_precision and world == 1:
        import subprocess
        osteps = max(1, min(args.steps, 3))
        cmd = [sys.executable, os.path.abspath(__file__), "--precision", str(1 - args.precision), "--one-precision", "--no-cpu-baseline",
               "--host-clips", "0", "--no-latency", "--no-clock", "--steps", str(osteps), "--warmup", "1", "--batch", str(B), "--height", str(H), "--width", str(W), "--encoder", args.encoder,
               "--flow-scale", str(args.flow_scale), "--flow-iters", str(args.flow_iters)] + (["--sequential-only"] if args.sequential_only else ["--seq-steps", "1"])
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0 and r.stdout.strip():
            o = json.loads(r.stdout.strip().splitlines()[-1])
            other = dict(o["this_precision"], steps=osteps, kernel_ms_per_step=o["kernel_ms_per_step"])
        else:
            other = {"error": (r.stderr or "")[-400:]}
    flow = flow_leg(args, R) if args.flow_pairs > 0 else None
    mask = mask_leg(args, R) if args.mask_frames > 0 else None
    gmf = gmflow_leg(args, R) if args.gmflow_pairs > 0 else None
    pipe = pipeline_leg(args, R) if args.pipeline_frames > 1 else None

    if rank == 0:
        dt, fam = main_res["dt"], main_res["fam"]
        fps = world * B * args.steps / dt
        # per-symbol attribution = the sequential pass (bands one after the other, every launch bracketed by HIP events): `sq` holds its
        # wall time, step count and per-band times; the timed region itself when --sequential-only
        sq = main_res["seq"]
        qsteps = sq["steps"]
        # roofline kernel: the family with the most launch time in the sequential pass (bands run one after the other, so this is
        # the kernel's own time - the criterion rocprofv3's per-symbol totals reproduce)
        # Two flow-band symbols sit within 1 % of each other, so a bare arg-max would name a different kernel from run to run: among the
        # symbols within 3 % of the largest time, the one with the most algorithmic FLOPs per step is taken (a property of the launch list).
        cands = [(k, v) for k, v in fam.items() if v["flops"] > 0]
        top_ms = max(v["ms"] for _, v in cands)
        dom_name, g = max(((k, v) for k, v in cands if v["ms"] >= 0.97 * top_ms), key=lambda kv: (kv[1]["flops"], kv[0]))
        ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        dom_sym = dom_name.split("/", 1)[1]
        dom_sym = SYMBOLS.get(dom_sym, dom_sym)
        traffic, traffic_src = pmc_traffic(dom_sym)
        exec_mult = g["exec_flops"] / g["flops"] if g["flops"] > 0 else 1.0     # MFMA passes issued per algorithmic pass (split-fp16: 2 - 3)
        tot_fl = sum(v["flops"] for v in fam.values())
        band_fl = {b: sum(v["flops"] for k, v in fam.items() if k.startswith(b + "/")) for b in ("depth", "flow")}

        def mode_summary(res, steps):
            return {"value": round(world * B * steps / res["dt"], 3), "unit": "frames/s", "ms_per_step": round(res["dt"] / steps * 1e3, 3),
                    "depth_ms_per_step": round(res["depth_s"] / steps * 1e3, 3), "flow_ms_per_step": round(res["flow_s"] / steps * 1e3, 3),
                    "band_ms_note": "ms from the step's start to the band's completion (in the default timed region both bands run at once)"}

        out = {
            "metric": "frames/sec (depth_anything ViT-L + flow_raft, 1080p)",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": PREC_NAME[args.precision], "data": "synthetic",
            "config": {"workload": f"every frame of a {B}-frame {W}x{H} synthetic uint8 clip (resident in HBM, one clip per GPU) through "
                                   f"depth_anything {args.encoder} (DINOv2 ViT-L/14 + DPT head, one batch of {B}, heat-encoded uint8 + min/max out) and "
                                   f"flow_raft (its {B - 1} consecutive forward pairs, --scale {args.flow_scale} -> {sw}x{sh}, {args.flow_iters} GRU iterations, "
                                   f"HSV-encoded uint8 + max displacement out); fused pre/post-process, seeded synthetic weights; "
                                   f"BASELINE.json configs[3] (depth, 32 frames per GPU) plus the flow band the metric names",
                       "frames_per_gpu_per_step": B, "frame": [H, W], "depth_net_input": list(engine.net_size(H, W)), "flow_net_input": [sh, sw],
                       "precision": PREC_NAME[args.precision],
                       "parallelism": f"one clip per GPU on {world} GPU(s); per-frame scalars all-gathered (12 bytes per frame), nothing else crosses GPUs"},
            "precision_modes": {
                "split-f16": "hi + lo operand pairs where the error budget needs them (the lo parts as fp16 or, on MX tiles, e4m3: 1.5-2 passes over K into one fp32 accumulator); max-norm and L2 error "
                             "< 1e-3 against every reference vector (range-relative: max |x - ref| / max |ref| and relative L2) - depth 5.5e-4 max / 3.2e-4 L2 on the reference's 720p vector, frames 0 / 13 / 31 of the 32 x 1080p batch this bench times "
                             "5.1e-4 / 7.0e-4 / 5.4e-4 max and the heavy-tailed weights on a 1080p frame 5.9e-4, all asserted below 7.5e-4 (tests/conftest.py MARGIN_DEPTH_SPLIT), "
                             "flow_raft 4.4e-4 ... 5.3e-4 max / 3.5e-4 L2 on 8 x 720p and 4.8e-4 / 3.7e-4 on 816 x 1440 against the reference (profiles/r06z_pytest_gpu_parity.log) - north_star's tolerance; the band scripts' mode.  "
                             "Pointwise (|x - ref| / max(|ref|, 1 % of the range), same log): median 2.6e-4 ... 2.9e-4 (depth) / 3.5e-4 (flow), 99.9th percentile 1.1e-2 ... 1.4e-2 / 2.1e-2 - "
                             "reached where |ref| is a few per cent of the range; the asserted bound is the range-relative one the min / max-normalised encodes see",
                "f16": "one fp16 MFMA pass per GEMM / conv, fp32 accumulate; against the fp32 reference depth 1.3e-3 max / 8e-4 L2, flow up to 2.2e-3 / 1.5e-3 "
                       "at 1280x720 - outside the tolerance, reported for comparison with round 1"},
            "roofline": {"bound": "mfma", "kernel": dom_sym, "family": dom_name,
                         "achieved": round(ach, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                         "algorithmic_bytes": round(g["bytes"] / max(g["launches"], 1)),
                         "avg_launch_ms": round(g["ms"] / max(g["launches"], 1), 5), "launches_per_step": g["launches"] / qsteps,
                         "flop_per_launch": g["flops"] / max(g["launches"], 1),
                         # the same symbol may serve both bands (the DPT head's and the update block's 256-channel convolutions): rocprofv3's
                         # per-symbol average is over all of them, so the line carries that figure too
                         "symbol_all_bands": (lambda same: {"launches_per_step": sum(v["launches"] for v in same) / qsteps,
                                                            "avg_launch_ms": round(sum(v["ms"] for v in same) / max(sum(v["launches"] for v in same), 1), 5)})(
                             [v for k, v in fam.items() if k.split("/", 1)[1] == dom_name.split("/", 1)[1]]),
                         "selection": ("kernel symbol with the largest summed launch time in the timed region" if args.sequential_only else
                                       "kernel symbol with the largest summed launch time in the labelled SEQUENTIAL pass (`sequential`: the same step with the bands one "
                                       "after the other, run right after the timed region; in the timed region both bands share the GPU and a launch's duration "
                                       "includes the other band's interference)") +
                                      " - HIP events on the band's stream around every launch; `family` = <band>/<symbol as rocprofv3 prints it>; symbols within "
                                      "3 % of the largest are ranked by their algorithmic FLOPs per step, so the name does not flip between runs; "
                                      "`rocprofv3 --kernel-trace --stats -- python bench.py --sequential-only` reproduces avg_launch_ms (profiles/)",
                         "step_frac": round(tot_fl / qsteps * args.steps / dt / 1e12 / PEAK_F16_TFLOPS, 4),
                         "depth_frac_alone": round(band_fl["depth"] / sq["depth_s"] / 1e12 / PEAK_F16_TFLOPS, 4),
                         "flow_frac_alone": round(band_fl["flow"] / sq["flow_s"] / 1e12 / PEAK_F16_TFLOPS, 4),
                         "executed_frac": round(ach * exec_mult / PEAK_F16_TFLOPS, 4),
                         "effective_clock_ghz": round(clock_ghz, 3) if clock_ghz else None,
                         "frac_at_clock": round(ach / (PEAK_F16_TFLOPS * clock_ghz / 2.4), 4) if clock_ghz else None,
                         "clock_note": "effective_clock_ghz = s_memtime cycles / s_memrealtime of the tiles of one fc1 + GELU launch of the ping-pong GEMM right after "
                                       "the timed region; frac_at_clock = achieved / (peak x clock / 2.4 GHz) - beside, not instead of, `frac` "
                                       "(profiles/r04a_clock_mfma_peak.txt: a bare MFMA loop holds 2.39 GHz on zeros and 1.63-1.65 GHz on random operands)",
                         "note": "flops are algorithmic multiply-adds (2 M N K of the layer; padding and the extra split-fp16 passes not counted) - executed_frac "
                                 "counts the MFMA work actually issued (x2 for weight-split, x3 for weight+activation-split layers); *_frac_alone = a "
                                 "band's flops / its wall time in the sequential pass / peak; step_frac = all launches' flops / timed step wall time / peak"},
            "model_tflops": round(tot_fl / qsteps * args.steps / dt / 1e12, 2),
            "executed_tflops": round(sum(v["exec_flops"] for v in fam.values()) / qsteps * args.steps / dt / 1e12, 2),
            "timed_region": ("sequential: depth band, then flow band, per-launch HIP events on (--sequential-only)" if args.sequential_only else
                             "both bands of a step enqueued at once on their two ctx streams (prisma_amd.engine.run_concurrently), no per-launch events; "
                             "results byte-identical to the sequential order (tests/test_gpu_edges.py)"),
            # the same step with the bands one after the other and every launch timed: what rounds 1-5 reported as `value`
            "sequential": {"value": round(world * B * qsteps / sq["dt"], 3), "unit": "frames/s", "steps": qsteps, "ms_per_step": round(sq["dt"] / qsteps * 1e3, 3),
                           "depth_ms_per_step": round(sq["depth_s"] / qsteps * 1e3, 3), "flow_ms_per_step": round(sq["flow_s"] / qsteps * 1e3, 3),
                           "kernel_ms_per_step_sum": round(sum(v["ms"] for v in fam.values()) / qsteps, 3),
                           "note": "per-launch HIP events cost ~2 % of this pass (rocprofv3 --stats sees the same launches without them)"},
            # socket power of this GPU over the timed region (hwmon power1_input sampled every 10 ms by prisma_amd/power.py): the part is power-limited
            # on this workload (profiles/r06_power_clock.txt), so joules per frame, not schedule slack, is what bounds frames/s
            "avg_power_w": main_res["power"]["avg_power_w"] if main_res.get("power") else None,
            "avg_sclk_mhz": main_res["power"]["avg_sclk_mhz"] if main_res.get("power") else None,
            "joules_per_frame": (round(main_res["power"]["avg_power_w"] * dt / (B * args.steps), 3)
                                 if main_res.get("power") and main_res["power"]["avg_power_w"] else None),
            "depth_anything": {"metric": f"frames/sec (depth_anything ViT-L, 1080p, batch {B} per GPU, inside the step)",
                               "value": round(world * B * qsteps / sq["depth_s"], 3), "unit": "frames/s",
                               "ms_per_step": round(sq["depth_s"] / qsteps * 1e3, 3),
                               "model_tflops": round(B * qsteps / sq["depth_s"] * GFLOP_PER_FRAME / 1e3, 2)},
            "flow_raft": {"metric": f"frame-pairs/sec (flow_raft, 1080p x {args.flow_scale}, {args.flow_iters} iterations, forward, inside the step)",
                          "value": round(world * (B - 1) * qsteps / sq["flow_s"], 3), "unit": "pairs/s",
                          "ms_per_step": round(sq["flow_s"] / qsteps * 1e3, 3)},
            "pcie_inclusive_fps": round(main_res["host_fps"], 2) if "host_fps" in main_res else None,
            "pcie_inclusive_note": "clips streamed through the asynchronous host-pointer entry points (pb_depth_submit_batch, pb_flow_submit_sequence, pb_wait) from page-locked "
                                   "frames into page-locked result arrays, the library's default chunking (a 32-frame clip is one chunk per band), "
                                   "two clips in flight per band so that a clip's uploads and downloads run under its neighbours' kernels, both bands at once like "
                                   "the timed region; results byte-identical to the HBM-resident leg's (--sequential-only: the blocking calls, one band after the other)",
            "latency_720p_batch1_ms": round(main_res["lat_b1"], 3) if "lat_b1" in main_res else None,
            "latency_720p_batch1_note": "one 1280x720 frame per call on a context created with max_batch = 1 (split-K on the launches with fewer tiles than CUs); "
                                        "the same call on the 32-frame context of the timed region, which never splits: "
                                        + ("%.3f ms" % main_res["lat_b1_big_ctx"] if "lat_b1_big_ctx" in main_res else "n/a"),
            "kernel_ms_per_step": {k: round(v["ms"] / qsteps, 3) for k, v in sorted(fam.items())},
            "kernel_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in fam.items()
                              if v["flops"] > 0 and v["ms"] > 0},
            "kernel_launches_per_step": {k: v["launches"] / qsteps for k, v in sorted(fam.items())},
            "kernel_effective_clock_ghz": pmc_clocks(sorted(fam))[0],
            "kernel_effective_clock_source": pmc_clocks(sorted(fam))[1],
            # algorithmic bytes (A + W + output once) per second of launch time: the figure to hold against HBM's ~8 TB/s for the launches
            # that move more than they compute (the K = 256 correlation-volume GEMMs sit in flow/gemm_kernel<128, 128, 2, 2, 0, 0, ...>)
            "kernel_algorithmic_tbps": {k: round(v["bytes"] / (v["ms"] * 1e-3) / 1e12, 3) for k, v in fam.items() if v["bytes"] > 0 and v["ms"] > 0},
        }
        out["this_precision"] = dict(mode_summary(main_res, args.steps), precision=PREC_NAME[args.precision])
        if other:
            out["other_precision"] = other
        if world == 1 and not args.no_cpu_baseline:
            cb = None
            if cpu_child is not None:
                try:
                    so, _ = cpu_child.communicate(timeout=600)
                    cb = json.loads(so.strip().splitlines()[-1])
                    cb["sample"] += "; run as a child process beside the GPU legs (host with >= 48 cores)"
                except Exception:      # noqa: BLE001 - fall back to the in-process baseline
                    cb = None
            out["cpu_baseline"] = cb or cpu_baseline(weights, cfg, rweights, frames, args.flow_scale, args.flow_iters)
        if flow:
            out["flow_raft_720p"] = flow
        if mask:
            out["mask_mmdet"] = mask
        if gmf:
            out["flow_gmflow"] = gmf
        if pipe:
            out["pipeline"] = pipe
        print(json.dumps(out))
    R.close()


if __name__ == "__main__":
    main()
