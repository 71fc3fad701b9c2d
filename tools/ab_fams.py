"""Per-family kernel time of the f16 and split modes for the library named by PRISMA_BANDS_LIB (A/B of two builds on one box):
python tools/ab_fams.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prisma_amd import engine, synth
B, H, W = 32, 1080, 1920
frames = torch.from_numpy(synth.frame_pair_sequence(B, H, W, seed=1000)).cuda()
rgb = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda")
sh, sw = engine.flow_out_size(H, W, 0.75)
frgb = torch.empty((B - 1, sh, sw, 3), dtype=torch.uint8, device="cuda")
sc = torch.zeros((3, B), dtype=torch.float32, device="cuda")
dw, rw = synth.depth_anything_weights("vitl", seed=1234), synth.raft_weights(seed=4321)
for prec in (0, 1):
    dn = engine.DepthAnything(dw, "vitl", max_batch=B, precision=prec)
    fn = engine.FlowRaft(rw, precision=prec)
    out = {}
    for name, net, call in (("depth", dn, lambda: dn.infer_dev(frames.data_ptr(), B, H, W, 0, rgb.data_ptr(), sc[0].data_ptr(), sc[1].data_ptr(), True)),
                            ("flow", fn, lambda: fn.infer_sequence_dev(frames.data_ptr(), B, H, W, 0.75, 12, False, 0, frgb.data_ptr(), sc[2].data_ptr()))):
        call(); net.sync()
        net.set_profiling(timing=True, accumulate=True)
        for _ in range(3):
            call(); net.sync()
        for s in net.kernel_stats():
            out[name + "/" + s["name"]] = round(s["ms"] / 3, 2)
        net.set_profiling(timing=False)
    print("precision", prec, os.environ.get("PRISMA_BANDS_LIB", "tree"), json.dumps(out), flush=True)
    dn.close(); fn.close()
