#!/usr/bin/env python3
"""Diagnostic: the packed-channel K axis (tile_n96 = 2) against the per-tap one (1) stage by stage, and both against the 125 x 157 golden vector."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from prisma_amd import engine, synth

def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
def rell2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))

z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "raft_125x157.npz"))
h, w = [int(v) for v in z["hw"]]
fr = synth.frame_pair_sequence(2, h, w, seed=int(z["frame_seed"]))
n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=1)
n.set_profiling(timing=False, debug_stages=True)
res = {}
for mode in (1, 2, 1, 2):
    n.set_option("tile_n96", mode)
    flow, _, _ = n.infer_sequence(fr, scale=1.0, iters=int(z["iters"]), backward=True)
    st = {k: n.stage(k).copy() for k in ("fmap", "net0", "corr0", "flow_it0", "flow_lo")}
    st["flow"] = flow.copy()
    if mode in res:
        print("mode", mode, "repeat identical:", all(np.array_equal(st[k], res[mode][k]) for k in st))
    res[mode] = st
    print("mode %d  fwd/golden relmax %.3e relL2 %.3e   bwd %.3e %.3e   fmap1/golden %.3e %.3e" % (
        mode, relmax(flow[0, 0], z["fwd"]), rell2(flow[0, 0], z["fwd"]), relmax(flow[0, 1], z["bwd"]), rell2(flow[0, 1], z["bwd"]),
        relmax(st["fmap"][:, ::4], z["fmap1"]), rell2(st["fmap"][:, ::4], z["fmap1"])))
for k in res[1]:
    d = np.abs(res[2][k].astype(np.float64) - res[1][k])
    print("%-9s mode 2 vs 1: relmax %.3e relL2 %.3e  share of elements that differ %.3f  shape %s" % (k, relmax(res[2][k], res[1][k]), rell2(res[2][k], res[1][k]), (d > 0).mean(), res[1][k].shape))
fm = np.abs(res[2]["fmap"].astype(np.float64) - res[1]["fmap"])
print("fmap diff per frame:", fm.reshape(fm.shape[0], -1).max(1), " per channel (max over 8):", np.sort(fm.max((0, 2, 3)))[-8:])
yy = fm.max((0, 1)); print("fmap diff by row:", np.round(yy.max(1) / (np.abs(res[1]["fmap"]).max()), 6))
n.close()
