import os, sys, subprocess
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from prisma_amd import engine, synth
z = np.load(os.path.join(%r, "tests", "golden", "raft_125x157.npz"))
h, w = [int(v) for v in z["hw"]]
fr = synth.frame_pair_sequence(2, h, w, seed=int(z["frame_seed"]))
n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=1)
n.set_profiling(timing=False, debug_stages=True)
n.set_option("tile_n96", int(sys.argv[2]))
flow, _, _ = n.infer_sequence(fr, scale=1.0, iters=int(z["iters"]), backward=True)
np.savez(sys.argv[1], fmap=n.stage("fmap"), net0=n.stage("net0"), flow=flow)
n.close()
''' % (root, root)
def run(tag, mode, env):
    f = "/tmp/cw_%s.npz" % tag
    subprocess.run([sys.executable, "-c", code, f, str(mode)], env=dict(os.environ, **env), check=True)
    return np.load(f)
def rell2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
base = run("base", 1, {})
for tag, mode, env in (("cw", 2, {}), ("tapin", 1, {"PB_TAPIN": "2"}), ("tapin_n96off", 0, {"PB_TAPIN": "2"}), ("halo_off", 1, {"PB_HALO": "0"}), ("cw_off_env", 2, {"PB_CW3": "0"})):
    r = run(tag, mode, env)
    print("%-14s vs per-tap 128x96: fmap relL2 %.3e  net0 %.3e  flow %.3e" % (tag, rell2(r["fmap"], base["fmap"]), rell2(r["net0"], base["net0"]), rell2(r["flow"], base["flow"])))
