"""Where does the fp16 error of the flow band come from?  CPU experiment on the pinned RAFT oracle: round classes of tensors
to fp16 (as the engine does) and measure the full-resolution flow against the exact fp32 oracle.
python tools/precision_budget_raft.py [H W iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as realF
from oracle import raft_oracle as RO
from prisma_amd import synth

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 320)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 12
w = synth.raft_weights(seed=4321)
fr = synth.frame_pair_sequence(2, H, W, seed=4)
flags = set()
group = ["enc"]
r16 = lambda t: t.half().float()

class FP:
    def __getattr__(self, n): return getattr(realF, n)
    def conv2d(self, x, wt, b=None, *a, **k):
        g = group[0]
        if ("A" + g) in flags: x = r16(x)
        if ("W" + g) in flags: wt = r16(wt)
        return realF.conv2d(x, wt, b, *a, **k)
    def grid_sample(self, c, g, **k):
        return realF.grid_sample(r16(c) if "VOL" in flags else c, g, **k)
RO.F = FP()
_enc, _upd, _pyr = RO.encoder, RO.update_block, RO.corr_pyramid
def enc(*a, **k):
    group[0] = "enc"; return _enc(*a, **k)
def upd(*a, **k):
    group[0] = "upd"; return _upd(*a, **k)
def pyr(f1, f2, levels=4):
    if "FM" in flags: f1, f2 = r16(f1), r16(f2)
    return _pyr(f1, f2, levels)
RO.encoder, RO.update_block, RO.corr_pyramid = enc, upd, pyr

ALL = ["Wenc", "Aenc", "Wupd", "Aupd", "FM", "VOL"]
def run(fl):
    flags.clear(); flags.update(fl)
    f, _ = RO.infer_pair(w, fr[0], fr[1], scale=1.0, iters=iters)
    return f
t0 = time.time(); ref = run([]); print("exact fp32: %.1f s, |flow| max %.3f" % (time.time() - t0, np.abs(ref).max()))
rng = float(np.abs(ref).max())
def row(name, fl):
    d = run(fl)
    print("%-40s relmax %.3e  relL2 %.3e" % (name, np.abs(d - ref).max() / rng, np.linalg.norm(d - ref) / np.linalg.norm(ref)), flush=True)
for f in ALL: row("only " + f, [f])
row("engine (all rounded)", ALL)
def without(*ex): return [f for f in ALL if f not in ex]
row("split W everywhere", without("Wenc", "Wupd"))
row("split W everywhere + A update block", without("Wenc", "Wupd", "Aupd"))
row("split W, A everywhere (volume fp16)", ["FM", "VOL"])
row("split W, A, fmaps (volume fp16)", ["VOL"])
