"""Error and cost of the split-fp16 variants on the GPU: ViT-L, one 1280x720 frame against the reference golden, and the time of
a 1080p batch.  PB_SPLIT=<vit_w><head_a><head_w> selects which operand classes are kept as hi + lo (engine.hip load()).
python tools/precision_modes.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prisma_amd import engine, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "depth_vitl_720p.npz"))
w = synth.depth_anything_weights("vitl", seed=1234)
frame = synth.frames(1, 720, 1280, seed=int(z["frame_seed"]))
clip = synth.frames(B, 1080, 1920, seed=3)
ref = z["depth_s8"].astype(np.float64)
rows = (("f16", 0, None, "1"), ("split 111, MX-fp8 ViT + head (default)", 1, "111", "1"), ("split 111, MX-fp8 ViT only", 1, "111", "10"),
        ("split 111, fp16 passes only", 1, "111", "0"),
        ("split 110", 1, "110", "0"), ("split 100", 1, "100", "0"), ("split 011", 1, "011", "0"), ("split 101", 1, "101", "0"))
if len(sys.argv) > 2: rows = rows[:4]
for name, prec, split, mx in rows:
    os.environ["PB_MX"] = mx
    if split: os.environ["PB_SPLIT"] = split
    else: os.environ.pop("PB_SPLIT", None)
    net = engine.DepthAnything(w, "vitl", device=0, max_batch=B, precision=prec)
    d = net.infer_batch(frame, want_rgb=False)[0][0][::8, ::8].astype(np.float64)
    emax, el2 = np.abs(d - ref).max() / np.abs(ref).max(), np.linalg.norm(d - ref) / np.linalg.norm(ref)
    net.infer_batch(clip, want_depth=False)
    t0 = time.perf_counter()
    net.infer_batch(clip, want_depth=False)
    dt = time.perf_counter() - t0
    net.set_profiling(timing=True)
    net.infer_batch(clip, want_depth=False)
    st = {s["name"]: round(s["ms"], 1) for s in net.kernel_stats()}
    print("%-40s relmax %.3e relL2 %.3e | %d x 1080p through the host API %.1f ms | kernel ms %s" % (name, emax, el2, B, dt * 1e3, st), flush=True)
    net.close()
