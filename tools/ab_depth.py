"""Depth band only: kernel ms per step of both precision modes under the environment given (A/B of PB_TAPIN / PB_MX settings on one box):
PB_TAPIN=0 python tools/ab_depth.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prisma_amd import engine, synth
B, H, W = 32, 1080, 1920
frames = torch.from_numpy(synth.frame_pair_sequence(B, H, W, seed=1000)).cuda()
rgb = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda")
sc = torch.zeros((2, B), dtype=torch.float32, device="cuda")
dw = synth.depth_anything_weights("vitl", seed=1234)
for prec in [int(p) for p in os.environ.get("AB_PREC", "0,1").split(",")]:
    dn = engine.DepthAnything(dw, "vitl", max_batch=B, precision=prec)
    call = lambda: dn.infer_dev(frames.data_ptr(), B, H, W, 0, rgb.data_ptr(), sc[0].data_ptr(), sc[1].data_ptr(), True)
    call(); dn.sync()
    dn.set_profiling(timing=True, accumulate=True)
    for _ in range(3):
        call(); dn.sync()
    out = {s["name"]: round(s["ms"] / 3, 2) for s in dn.kernel_stats()}
    dn.set_profiling(timing=False)
    tag = " ".join(f"{k}={os.environ[k]}" for k in ("PB_TAPIN", "PB_MX") if k in os.environ) or "default"
    print(f"[{tag}] precision {prec} total {sum(out.values()):.1f} ms", json.dumps(out), flush=True)
    dn.close()
