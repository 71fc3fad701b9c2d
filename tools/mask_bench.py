"""Stand-alone timing of the mask_mmdet leg (for rocprofv3): python tools/mask_bench.py [frames] [steps]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from prisma_amd import engine, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = synth.MASK_CFGS["r101"]
net = engine.MaskMMDet(synth.solov2_weights(cfg), cfg, max_batch=min(B, int(os.environ.get("MASK_CHUNK", "8"))))
frames = torch.from_numpy(synth.frames(B, 1080, 1920, seed=70)).cuda()
out = torch.empty_like(frames)
keep = [synth.COCO_CLASSES.index(c) for c in synth.BAND_CLASSES]
net.infer_batch_dev(frames.data_ptr(), B, 1080, 1920, 0.5, keep, out.data_ptr()); net.sync()
prof = os.environ.get('MASK_PROF', '1') == '1'
net.set_profiling(prof)
t0 = time.perf_counter()
for _ in range(steps):
    net.infer_batch_dev(frames.data_ptr(), B, 1080, 1920, 0.5, keep, out.data_ptr()); net.sync()
dt = (time.perf_counter() - t0) / steps
st = net.kernel_stats() if prof else []
print("frames/s %.1f  ms/step %.2f  kernel ms %s" % (B / dt, dt * 1e3, {s["name"]: round(s["ms"], 2) for s in st}))
