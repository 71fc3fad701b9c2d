"""Where does the fp16 error of the depth band come from?  CPU experiment on the pinned oracle: round classes of tensors to
fp16 (as the engine does) and measure the final relative-depth error against the exact fp32 oracle.  Rows: one class at a
time, then candidate precision modes (= everything rounded EXCEPT the classes a split-fp16 pass would restore).
python tools/precision_budget.py [H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as realF
from oracle import depth_oracle as O
from prisma_amd import synth

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (360, 640)
w = synth.depth_anything_weights("vitl", seed=1234)
fr = synth.frames(1, H, W, seed=0)[0]
x = O.preprocess(fr)[None] if O.preprocess(fr).ndim == 3 else O.preprocess(fr)
flags = set()
r16 = lambda t: t.half().float()
HI = 200          # head maps at least this wide are "high resolution" (refinenet1, output convs): 3/4 of the head's FLOPs

class FP:
    def __getattr__(self, n): return getattr(realF, n)
    def linear(self, x, wt, b=None):
        return realF.linear(r16(x) if "Avit" in flags else x, r16(wt) if "Wvit" in flags else wt, b)
    def conv2d(self, x, wt, b=None, *a, **k):
        if wt.shape[-1] == 14:
            return realF.conv2d(r16(x) if "PATCH" in flags else x, r16(wt) if "Wvit" in flags else wt, b, *a, **k)
        hi = x.shape[-1] >= HI
        if ("AheadHi" if hi else "AheadLo") in flags: x = r16(x)
        if ("WheadHi" if hi else "WheadLo") in flags: wt = r16(wt)
        return realF.conv2d(x, wt, b, *a, **k)
    def conv_transpose2d(self, x, wt, b=None, *a, **k):
        return realF.conv_transpose2d(r16(x) if "AheadLo" in flags else x, r16(wt) if "WheadLo" in flags else wt, b, *a, **k)
O.F = FP()
_mm = torch.Tensor.__matmul__
def mm(a, b):
    if "QKV" in flags: a, b = r16(a), r16(b)
    return _mm(a, b)
torch.Tensor.__matmul__ = mm
_sm = torch.Tensor.softmax
def sm(t, *a, **k):
    o = _sm(t, *a, **k)
    return r16(o) if "P" in flags else o
torch.Tensor.softmax = sm

ALL = ["Wvit", "WheadLo", "WheadHi", "Avit", "AheadLo", "AheadHi", "QKV", "P", "PATCH"]
def run(fl):
    flags.clear(); flags.update(fl)
    return O.model_forward(w, x, depth=24, heads=16)
t0 = time.time(); ref = run([]); print("exact fp32: %.1f s, depth range %.3f .. %.3f" % (time.time() - t0, ref.min(), ref.max()))
rng = float(np.abs(ref).max())
def row(name, fl):
    d = run(fl)
    print("%-44s relmax %.3e  relL2 %.3e" % (name, np.abs(d - ref).max() / rng, np.linalg.norm(d - ref) / np.linalg.norm(ref)), flush=True)
for f in ALL:
    row("only " + f, [f])
row("engine (all rounded)", ALL)
def without(*ex): return [f for f in ALL if f not in ex]
row("split W everywhere", without("Wvit", "WheadLo", "WheadHi"))
row("split W everywhere + A head", without("Wvit", "WheadLo", "WheadHi", "AheadLo", "AheadHi"))
row("split W + A in the head only", without("WheadLo", "WheadHi", "AheadLo", "AheadHi"))
row("split W + A in the low-res head only", without("WheadLo", "AheadLo"))
row("split W vit + W/A low-res head", without("Wvit", "WheadLo", "AheadLo"))
row("split W vit + W/A head", without("Wvit", "WheadLo", "AheadLo", "WheadHi", "AheadHi"))
row("split W,A everywhere (3-pass)", ["QKV", "P"])
