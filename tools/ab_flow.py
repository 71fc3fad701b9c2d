"""Flow band only: kernel ms per step of both precision modes under the environment given (A/B of PB_TAPIN / PB_MX settings on one box):
PB_TAPIN=0 python tools/ab_flow.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prisma_amd import engine, synth
B, H, W = int(os.environ.get("AB_FRAMES", "32")), 1080, 1920
frames = torch.from_numpy(synth.frame_pair_sequence(B, H, W, seed=1000)).cuda()
sh, sw = engine.flow_out_size(H, W, 0.75)
frgb = torch.empty((B - 1, sh, sw, 3), dtype=torch.uint8, device="cuda")
sc = torch.zeros((B,), dtype=torch.float32, device="cuda")
rw = synth.raft_weights(seed=4321)
for prec in [int(p) for p in os.environ.get("AB_PREC", "0,1").split(",")]:
    fn = engine.FlowRaft(rw, precision=prec)
    call = lambda: fn.infer_sequence_dev(frames.data_ptr(), B, H, W, 0.75, 12, False, 0, frgb.data_ptr(), sc.data_ptr())
    call(); fn.sync()
    fn.set_profiling(timing=True, accumulate=True)
    for _ in range(3):
        call(); fn.sync()
    out = {s["name"]: round(s["ms"] / 3, 2) for s in fn.kernel_stats()}
    fn.set_profiling(timing=False)
    tag = " ".join(f"{k}={os.environ[k]}" for k in ("PB_TAPIN", "PB_MX", "PB_MX_UPD") if k in os.environ) or "default"
    print(f"[{tag}] precision {prec} total {sum(out.values()):.1f} ms", json.dumps(out), flush=True)
    fn.close()
