import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

OP_CASES = ["testpy_double", "testpy_float", "cfg1", "cfg2_sub", "oob", "edges", "chunk"] + \
           ["testpy_grad_D%d" % d for d in (30, 32, 64, 71, 1025, 2048, 3096)]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def rel_err(got, ref):
    """max |got - ref| relative to max |ref| (the tensors here have no natural per-element scale)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-300)) if ref.size else 0.0


def near_boundary_mask(z, tol=1e-4):
    """Points of a golden case whose pixel coordinate is within `tol` of an integer.  The gradient
    w.r.t. the location jumps there (bilinear interpolation has a kink at every pixel centre, and
    at -1 / W / H the point enters or leaves the map), so an fp32 and an fp64 evaluation of
    loc*W-0.5 may legitimately land on different sides; forward values and the other two
    gradients are continuous and are compared everywhere."""
    shapes = z["shapes"].astype(np.float64)
    wh = np.stack([shapes[:, 1], shapes[:, 0]], -1)[None, None, None, :, None, :]
    pix = z["loc"].astype(np.float64) * wh - 0.5
    near = np.abs(pix - np.round(pix)) < tol
    return near.any(-1)


@pytest.fixture(scope="session")
def oracle():
    from oracle import msda_oracle
    msda_oracle.build()
    return msda_oracle
