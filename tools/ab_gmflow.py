"""flow_gmflow band only: kernel ms per step (15 forward pairs of a 1080p clip at --scale 0.75) under the environment given, for same-box A/B runs:
python tools/ab_gmflow.py"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prisma_amd import engine, synth
P, H, W = int(os.environ.get("AB_PAIRS", "15")), 1080, 1920
frames = torch.from_numpy(synth.frame_pair_sequence(P + 1, H, W, seed=150)).cuda()
sh, sw = engine.flow_out_size(H, W, 0.75)
rgb = torch.empty((P, sh, sw, 3), dtype=torch.uint8, device="cuda")
mx = torch.zeros((P,), dtype=torch.float32, device="cuda")
net = engine.FlowGMFlow(synth.gmflow_weights(seed=2468))
call = lambda: net.infer_sequence_dev(frames.data_ptr(), P + 1, H, W, 0.75, 1, False, 0, rgb.data_ptr(), mx.data_ptr())
call(); net.sync()
t0 = time.perf_counter()
for _ in range(3):
    call(); net.sync()
dt = (time.perf_counter() - t0) / 3
net.set_profiling(timing=True, accumulate=True)
for _ in range(2):
    call(); net.sync()
out = {s["name"]: round(s["ms"] / 2, 2) for s in net.kernel_stats()}
print(f"pairs/s {P / dt:.1f}  ms/step {dt * 1e3:.1f}  kernels", json.dumps(out), flush=True)
net.close()
