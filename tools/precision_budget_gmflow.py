"""Where would the fp16 error of a flow_gmflow engine come from?  CPU experiment on the pinned GMFlow oracle (the HIP path of this band is
not built yet: this sizes the precision modes it will need): round one class of operands to fp16 and measure the full-resolution flow
against the exact fp32 oracle.
python tools/precision_budget_gmflow.py [H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as realF
from oracle import gmflow_oracle as G
from prisma_amd import synth

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (216, 300)
torch.set_num_threads(16)
w = synth.gmflow_weights(seed=2468)
fr = synth.frame_pair_sequence(2, H, W, seed=52)
flags, ctx = set(), ["bb"]
r16 = lambda t: t.half().float()


class FP:
    def __getattr__(self, n):
        return getattr(realF, n)

    def conv2d(self, x, wt, b=None, *a, **k):
        g = ctx[0] if ctx[0] in ("bb", "up") else "bb"
        if ("A" + g) in flags: x = r16(x)
        if ("W" + g) in flags: wt = r16(wt)
        return realF.conv2d(x, wt, b, *a, **k)

    def linear(self, x, wt, b=None):
        g = "prop" if ctx[0] == "prop" else "lin"
        if ("A" + g) in flags: x = r16(x)
        if ("W" + g) in flags: wt = r16(wt)
        return realF.linear(x, wt, b)


class TP:
    """torch proxy: matmul rounds its operands per context (attention scores / P V, matching correlation / expectation, propagation)."""
    def __getattr__(self, n):
        return getattr(torch, n)

    def matmul(self, a, b):
        c = ctx[0]
        first = b.shape[-1] != 2 and not getattr(a, "_is_prob", False)        # scores: [.., L, C] x [.., C, L]
        key = ("S" if first else "P") + c
        if key in flags: a, b = r16(a), r16(b)
        return torch.matmul(a, b)

    def softmax(self, x, dim):
        p = torch.softmax(x, dim)
        p._is_prob = True
        return p


G.F, G.torch = FP(), TP()
_wa, _gc, _fa, _up, _bb = G.window_attention, G.global_correlation_softmax, G.flow_attention, G.upsample_flow, G.backbone
def wrap(fn, name):
    def f(*a, **k):
        old = ctx[0]; ctx[0] = name
        try: return fn(*a, **k)
        finally: ctx[0] = old
    return f
G.window_attention, G.global_correlation_softmax, G.flow_attention = wrap(_wa, "attn"), wrap(_gc, "match"), wrap(_fa, "prop")
G.upsample_flow, G.backbone = wrap(_up, "up"), wrap(_bb, "bb")
_softmax = realF.softmax
def fsoftmax(x, dim=-1):
    p = _softmax(x, dim=dim); p._is_prob = True; return p
FP.softmax = staticmethod(fsoftmax)

ALL = [("Wbb", "backbone conv weights"), ("Abb", "backbone conv inputs"), ("Wlin", "transformer linear weights"), ("Alin", "transformer linear inputs"),
       ("Sattn", "window attention: q, k"), ("Pattn", "window attention: softmax P, v"), ("Smatch", "matching: features into the 18 360-way correlation"),
       ("Pmatch", "matching: probabilities x coordinates"), ("Wprop", "propagation: q / k projection weights"), ("Aprop", "propagation: projection inputs"),
       ("Sprop", "propagation: q, k"), ("Pprop", "propagation: probabilities x flow"), ("Wup", "upsampler conv weights"), ("Aup", "upsampler conv inputs")]


def run(fl):
    flags.clear(); flags.update(fl); ctx[0] = "lin"
    f, _ = G.infer_pair(w, fr[0], fr[1], scale=1.0, backward=False)
    return f


t0 = time.time(); ref = run([]); print("exact fp32: %.1f s, |flow| max %.2f px at %dx%d" % (time.time() - t0, np.abs(ref).max(), W, H))
rng = float(np.abs(ref).max())
print("| class rounded to fp16 | max/range | rel L2 |\n|---|---|---|")
for k, name in ALL:
    d = run([k])
    print("| %s | %.1e | %.1e |" % (name, np.abs(d - ref).max() / rng, np.linalg.norm(d - ref) / np.linalg.norm(ref)), flush=True)
d = run([k for k, _ in ALL])
print("| **all of them** | **%.1e** | **%.1e** |" % (np.abs(d - ref).max() / rng, np.linalg.norm(d - ref) / np.linalg.norm(ref)))
d = run([k for k, _ in ALL if k[0] == "W"])
print("| all weights | %.1e | %.1e |" % (np.abs(d - ref).max() / rng, np.linalg.norm(d - ref) / np.linalg.norm(ref)))
