"""Which ViT linears need their weight residual?  CPU experiment on the pinned depth oracle (ViT-L): everything the split mode leaves in
single fp16 is rounded (GEMM inputs, q / k / v, softmax P, patches), the DPT head is exact (its operands are split), and the weights of a
SUBSET of the 96 ViT linears are rounded to fp16 as well - the subset whose e4m3 residual segment the engine would skip.
python tools/precision_budget_vit_layers.py [H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as realF
from oracle import depth_oracle as O
from prisma_amd import synth

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (360, 640)
w = synth.depth_anything_weights("vitl", seed=1234)
fr = synth.frames(1, H, W, seed=0)[0]
x = O.preprocess(fr)[None]
r16 = lambda t: t.half().float()
KIND = {(3072, 1024): "qkv", (1024, 1024): "proj", (4096, 1024): "fc1", (1024, 4096): "fc2"}
state = {"skip": lambda kind, blk: False, "count": {}}

class FP:
    def __getattr__(self, n): return getattr(realF, n)
    def linear(self, x, wt, b=None):
        kind = KIND.get(tuple(wt.shape))
        if kind is None:
            return realF.linear(x, wt, b)
        blk = state["count"].get(kind, 0)
        state["count"][kind] = blk + 1
        return realF.linear(r16(x), r16(wt) if state["skip"](kind, blk) else wt, b)
    def conv2d(self, x, wt, b=None, *a, **k):
        if wt.shape[-1] == 14:
            return realF.conv2d(r16(x), wt, b, *a, **k)
        return realF.conv2d(x, wt, b, *a, **k)
O.F = FP()
_mm = torch.Tensor.__matmul__
torch.Tensor.__matmul__ = lambda a, b: _mm(r16(a), r16(b))
_sm = torch.Tensor.softmax
torch.Tensor.softmax = lambda t, *a, **k: r16(_sm(t, *a, **k))

def run(skip):
    state["skip"], state["count"] = skip, {}
    return O.model_forward(w, x, depth=24, heads=16)

# exact reference: no rounding at all
O.F = realF; torch.Tensor.__matmul__ = _mm; torch.Tensor.softmax = _sm
ref = O.model_forward(w, x, depth=24, heads=16)
O.F = FP(); torch.Tensor.__matmul__ = lambda a, b: _mm(r16(a), r16(b)); torch.Tensor.softmax = lambda t, *a, **k: r16(_sm(t, *a, **k))
rng = float(np.abs(ref).max())
def row(name, skip):
    t0 = time.time()
    d = run(skip)
    print("%-58s relmax %.3e  relL2 %.3e   (%.0f s)" % (name, np.abs(d - ref).max() / rng, np.linalg.norm(d - ref) / np.linalg.norm(ref), time.time() - t0), flush=True)
row("split mode as shipped (no weight rounded)", lambda k, b: False)
for kind in ("qkv", "proj", "fc1", "fc2"):
    row("residual skipped on every " + kind, lambda k, b, kind=kind: k == kind)
row("skipped on blocks 0-11", lambda k, b: b < 12)
row("skipped on blocks 12-23", lambda k, b: b >= 12)
row("skipped on fc1 + fc2", lambda k, b: k in ("fc1", "fc2"))
row("skipped on fc1 + fc2 of blocks 0-11", lambda k, b: k in ("fc1", "fc2") and b < 12)
row("skipped everywhere (= weights single fp16)", lambda k, b: True)
