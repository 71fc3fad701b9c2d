#!/usr/bin/env python3
"""How much of the step can stream-level overlap buy?  The bench's 32-frame 1080p clip through depth_anything + flow_raft in several
enqueue layouts - every layout runs exactly the work of one bench step and produces the same bytes:

  seq        depth(32) then flow(31 pairs), one after the other (the round-5 timed region)
  2way       depth(32) | flow(31 pairs) on their two ctx streams, enqueued from one host thread
  2way-thr   the same from two host threads
  3way       depth(32) | flow(pairs 0..15) | flow(pairs 16..30): two flow contexts on sub-clips sharing one halo frame
  4way       depth(0..15) | depth(16..31) | flow(0..15) | flow(16..30)
  ...-thr    one host thread per context
  flow1 / flow2, depth1 / depth2   one band alone: whole clip on one context / two half clips on two contexts
  2way-xcd   2way with the depth stream on XCDs 0-3 and the flow stream on XCDs 4-7 (PB_CU_MASK_DEPTH / PB_CU_MASK_FLOW, engine.h pb_create_stream)
  2way-cu    2way with every XCD's CUs dealt alternately to the two streams
  2way-fprio / 2way-dprio / 2way-flow-low   2way with the flow / depth stream on a high-priority queue, the flow stream on a low-priority one

python tools/overlap_bench.py [--steps 6] [--layouts seq,2way,...]"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from prisma_amd.power import PowerSampler  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--layouts", default="seq,2way,2way-thr,3way,3way-thr,4way,4way-thr")
    ap.add_argument("--precision", type=int, default=1)
    args = ap.parse_args()
    from prisma_amd import engine, synth
    B, H, W = args.batch, 1080, 1920
    cfg = synth.DEPTH_CFGS["vitl"]
    weights = synth.cached_weights("depth", cfg, 1234)
    rweights = synth.cached_weights("raft", 4321)
    frames = synth.frame_pair_sequence(B, H, W, seed=1000)
    d_frames = torch.from_numpy(frames).cuda()
    sh, sw = engine.flow_out_size(H, W, 0.75)
    fbytes = H * W * 3

    def outputs():
        return (torch.zeros((B, H, W, 3), dtype=torch.uint8, device="cuda"), torch.zeros((B - 1, sh, sw, 3), dtype=torch.uint8, device="cuda"),
                torch.zeros((3, B), dtype=torch.float32, device="cuda"))

    ref = None
    results = {}
    for layout in args.layouts.split(","):
        base = layout.replace("-thr", "")
        threaded = layout.endswith("-thr")
        nd = {"4way": 2, "depth2": 2, "flow1": 0, "flow2": 0}.get(base, 1)
        nf = {"3way": 2, "4way": 2, "flow2": 2, "depth1": 0, "depth2": 0}.get(base, 1)
        for k in ("PB_CU_MASK_DEPTH", "PB_CU_MASK_FLOW", "PB_CU_MASK_DEPTH_PRIO", "PB_CU_MASK_FLOW_PRIO"):
            os.environ.pop(k, None)
        if base == "2way-fprio":
            os.environ["PB_CU_MASK_FLOW_PRIO"] = "-1"
        if base == "2way-dprio":
            os.environ["PB_CU_MASK_DEPTH_PRIO"] = "-1"
        if base == "2way-flow-low":
            os.environ["PB_CU_MASK_FLOW_PRIO"] = "1"
        if base == "2way-xcd":
            os.environ["PB_CU_MASK_DEPTH"] = ",".join(["0f0f0f0f"] * 8); os.environ["PB_CU_MASK_FLOW"] = ",".join(["f0f0f0f0"] * 8)
        if base == "2way-cu":
            os.environ["PB_CU_MASK_DEPTH"] = ",".join(["00ff00ff"] * 8); os.environ["PB_CU_MASK_FLOW"] = ",".join(["ff00ff00"] * 8)
        d_rgb, f_rgb, scal = outputs()
        dsplit = [(i * B // nd, (i + 1) * B // nd) for i in range(nd)]                  # frame ranges
        P = B - 1
        fsplit = [(i * ((P + nf - 1) // nf), min(P, (i + 1) * ((P + nf - 1) // nf))) for i in range(nf)]      # pair ranges
        if base in ("depth1", "depth2", "flow1", "flow2"):
            base = "par"
        dns = [engine.DepthAnything(weights, cfg, device=0, max_batch=b - a, precision=args.precision) for a, b in dsplit]
        fns = [engine.FlowRaft(rweights, device=0, precision=args.precision) for _ in fsplit]
        jobs = []
        for net, (a, b) in zip(dns, dsplit):
            def job(net=net, a=a, b=b):
                net.infer_dev(d_frames.data_ptr() + a * fbytes, b - a, H, W, 0, d_rgb[a:].data_ptr(), scal[0, a:].data_ptr(), scal[1, a:].data_ptr(), True)
            jobs.append((net, job))
        for net, (a, b) in zip(fns, fsplit):
            def job(net=net, a=a, b=b):
                net.infer_sequence_dev(d_frames.data_ptr() + a * fbytes, b - a + 1, H, W, 0.75, 12, False, 0, f_rgb[a:].data_ptr(), scal[2, a:].data_ptr())
            jobs.append((net, job))

        done_ms = [0.0] * len(jobs)

        def step():
            if base == "seq":
                for net, job in jobs:
                    job(); net.sync()
            elif threaded:
                def run(net, job):
                    job(); net.sync()
                ths = [threading.Thread(target=run, args=j) for j in jobs[1:]]
                for t in ths:
                    t.start()
                run(*jobs[0])
                for t in ths:
                    t.join()
            else:
                t_ = time.perf_counter()
                for net, job in jobs:
                    job()
                for i, (net, job) in enumerate(jobs):
                    net.sync()
                    done_ms[i] += (time.perf_counter() - t_) * 1e3

        step(); step()
        torch.cuda.synchronize()
        for i in range(len(done_ms)):
            done_ms[i] = 0.0
        with PowerSampler() as ps:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        dt = (t1 - t0) / args.steps
        pw = ps.window(t0 + 0.1 * (t1 - t0), t1)
        out = (d_rgb.cpu().numpy(), f_rgb.cpu().numpy(), scal.cpu().numpy())
        same = None
        if ref is None and nd and nf:
            ref = out
        elif ref is not None:
            same = all(np.array_equal(x, y) for x, y, on in zip(ref, out, (nd, nf, 0)) if on) and (not nd or np.array_equal(ref[2][:2], out[2][:2])) \
                and (not nf or np.array_equal(ref[2][2], out[2][2]))
        results[layout] = {"ms_per_step": round(dt * 1e3, 2), "fps": round(B / dt, 2), "bytes_equal_to_first": same, **pw,
                           "joules_per_frame": round(pw["avg_power_w"] * dt / B, 2) if pw["avg_power_w"] else None,
                           "ctx_done_ms": [round(x / args.steps, 1) for x in done_ms] if any(done_ms) else None}
        print(layout, results[layout], flush=True)
        for n in dns + fns:
            n.close()
        del d_rgb, f_rgb, scal
        torch.cuda.empty_cache()
    print(json.dumps(results))


if __name__ == "__main__":
    main()
