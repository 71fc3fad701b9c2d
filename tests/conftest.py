import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLD


# north_star: "within 1e-3 relative on float depth/flow".  Metric: max|x - ref| / max|ref| and relative L2.
# PB_PREC_SPLIT (the engine classes' default) is the mode that bound is asserted in; PB_PREC_F16 (single fp16 pass, the faster
# mode bench.py also reports) is held to what an 11-bit-mantissa operand rounding delivers on these models.
TOL = {1: (1e-3, 1e-3),       # precision -> (max / range, L2)
       0: (2e-3, 1.1e-3)}
# The depth band's split mode runs qkv and fc1 without a weight-residual pass (round 3: the per-layer assignment bought 10 ms of 150).
# What that left of the 1e-3 is a TESTED quantity: at the bench's own shape - three frames of the 32 x 1080p batch
# against the oracle, and the heavy-tailed weights on a 1080p frame against the real reference - the max-norm error stays below this.
MARGIN_DEPTH_SPLIT = 7.5e-4



def pointwise(a, b, floor=0.01):
    """a POINTWISE figure beside the range-relative one - |x - ref| / max(|ref|, floor x range(ref)) per element, as
    (median, 99.9th percentile, max).  range = max - min for a scalar map, max |ref| for a vector field (last axis 2).  Reported in the
    parity log next to relmax / relL2; the asserted tolerances stay the range-relative ones the encodes are normalised by (README)."""
    import numpy as np
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    rng = float(np.abs(b).max()) if b.ndim >= 3 and b.shape[-1] == 2 else float(b.max() - b.min())
    e = np.abs(a - b) / np.maximum(np.abs(b), floor * rng + 1e-30)
    return float(np.median(e)), float(np.percentile(e, 99.9)), float(e.max())


def pw(a, b):
    return "pointwise p50 %.1e p99.9 %.1e max %.1e" % pointwise(a, b)


# the band scripts refuse to run without a checkpoint unless seeded synthetic weights are asked for; tests ask
os.environ.setdefault("PRISMA_SYNTH", "1")
