"""Which update-block layers need their weights split?  The `Wupd` row of tools/precision_budget_raft.py broken down by sub-module:
round the WEIGHTS of one sub-module to fp16 (everything else exact) and measure the full-resolution flow against the fp32 oracle.
python tools/precision_budget_raft_fine.py [H W iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as realF
from oracle import raft_oracle as RO
from prisma_amd import synth

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 320)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 12
torch.set_num_threads(16)
w = {k: np.ascontiguousarray(v) for k, v in synth.raft_weights(seed=4321).items()}
name_of = {v.ctypes.data: k for k, v in w.items() if v.ndim == 4}
fr = synth.frame_pair_sequence(2, H, W, seed=4)
GROUPS = {"motion encoder (convc1, convc2, convf1, convf2, conv)": "update_block.encoder.", "GRU z / r / q convs": "update_block.gru.",
          "flow head": "update_block.flow_head.", "mask head": "update_block.mask.", "fnet": "fnet.", "cnet": "cnet."}
active = [None]
r16 = lambda t: t.half().float()


class FP:
    def __getattr__(self, n):
        return getattr(realF, n)

    def conv2d(self, x, wt, b=None, *a, **k):
        n = name_of.get(wt.data_ptr(), "")
        if active[0] and n.startswith(active[0]):
            wt = r16(wt)
        return realF.conv2d(x, wt, b, *a, **k)


RO.F = FP()


def run(prefix):
    active[0] = prefix
    f, _ = RO.infer_pair(w, fr[0], fr[1], scale=1.0, iters=iters)
    return f


t0 = time.time(); ref = run(None); print("exact fp32: %.1f s, |flow| max %.3f" % (time.time() - t0, np.abs(ref).max()))
rng = float(np.abs(ref).max())
for name, prefix in GROUPS.items():
    d = run(prefix)
    print("weights of %-56s relmax %.3e  relL2 %.3e" % (name, np.abs(d - ref).max() / rng, np.linalg.norm(d - ref) / np.linalg.norm(ref)), flush=True)
d = run("update_block.")
print("weights of %-56s relmax %.3e  relL2 %.3e" % ("the whole update block", np.abs(d - ref).max() / rng, np.linalg.norm(d - ref) / np.linalg.norm(ref)))
