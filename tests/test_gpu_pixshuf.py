"""The buffer-addressed pixel-shuffle epilogue (gemm_kernels.h pixshuf_epilogue_buf: one division per wave, scalar row offsets, counted grid-row
wraps) must store the very bits of the flat-addressed one it replaces (PB_PIXSHUF_BUF=0, read once per process): the flow band's encoder
stem (3 x 3 convolution over the space-to-depth frame, s = 2, 64 channels) and the depth band's reassemble stage (transposed convolutions
as 1 x 1 GEMMs, s = 4 and s = 2, 256 / 512 channels), in both precision modes (plain fp16 maps / split maps with e4m3 residual parts).
Grids narrower than a wave tile's 128 rows (46 and 33 columns here) make a wave's rows wrap up to three times."""
import os
import subprocess
import sys

import numpy as np
import pytest

from prisma_amd import engine, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flat(code, out):
    r = subprocess.run([sys.executable, "-c", "import sys, numpy as np; sys.path.insert(0, %r); from prisma_amd import engine, synth; " % ROOT + code],
                       env=dict(os.environ, PB_PIXSHUF_BUF="0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-800:]
    return np.load(out)


@pytest.mark.parametrize("prec", [1, 0])
def test_flow_stem_same_bits_as_flat_epilogue(prec, tmp_path):
    fr = synth.frame_pair_sequence(3, 131, 181, seed=33)
    n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=prec)
    flow, _, _ = n.infer_sequence(fr, scale=1.0, iters=4, backward=True)
    n.close()
    out = str(tmp_path / "flat.npy")
    ref = _flat("fr = synth.frame_pair_sequence(3, 131, 181, seed=33); n = engine.FlowRaft(synth.raft_weights(seed=4321), device=0, precision=%d); "
                "f, _, _ = n.infer_sequence(fr, scale=1.0, iters=4, backward=True); np.save(%r, f)" % (prec, out), out)
    assert np.array_equal(ref, flow)


@pytest.mark.parametrize("prec", [1, 0])
def test_depth_reassemble_same_bits_as_flat_epilogue(prec, tmp_path):
    c = synth.DEPTH_CFGS["vitl_d4"]
    frames = synth.frames(2, 266, 462, seed=5)                  # 19 x 33 patch grid
    net = engine.DepthAnything(synth.depth_anything_weights(c, seed=1234), c, device=0, max_batch=2, precision=prec)
    depth, _, _, _ = net.infer_batch(frames)
    net.close()
    out = str(tmp_path / "flat.npy")
    ref = _flat("c = synth.DEPTH_CFGS['vitl_d4']; fr = synth.frames(2, 266, 462, seed=5); "
                "net = engine.DepthAnything(synth.depth_anything_weights(c, seed=1234), c, device=0, max_batch=2, precision=%d); "
                "d, _, _, _ = net.infer_batch(fr); np.save(%r, d)" % (prec, out), out)
    assert np.array_equal(ref, depth)
