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

# the band scripts refuse to run without a checkpoint unless seeded synthetic weights are asked for; tests ask
os.environ.setdefault("PRISMA_SYNTH", "1")
