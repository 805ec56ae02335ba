import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name)))
        return cache[name]

    return load


def rel_l2(a, b):
    """||a-b|| / ||b|| in float64."""
    import torch
    a = torch.as_tensor(np.asarray(a) if not hasattr(a, "detach") else a.detach().cpu().numpy()).double()
    b = torch.as_tensor(np.asarray(b) if not hasattr(b, "detach") else b.detach().cpu().numpy()).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def grad_close(got, ref32, ref64):
    """Mask-flip-robust gradient criterion for deep ReLU / LeakyReLU nets.  A pre-activation within rounding of zero
    flips its mask, and one flipped element among millions moves a gradient by ~1e-3 rel-L2; whether the reference's
    own fp32 evaluation happens to flip one is luck.  Both parts are against the fp64 evaluation of the reference:
    rel-L2 <= max(5e-3, 3 * err(reference fp32)) -- a wrong slope, stride or tap order is off by > 1e-1 -- and the 90th
    percentile of |difference| <= 1e-4 * rms(reference): outside the few flipped receptive fields the agreement is at
    fp32 rounding level.  Returns (ok, (err, err_ref32, p90/rms))."""
    import torch
    got, ref32, ref64 = (t.detach().cpu().double() for t in (got, ref32, ref64))
    e_got, e_ref = rel_l2(got, ref64), rel_l2(ref32, ref64)
    rms = float(ref64.pow(2).mean().sqrt())
    p90 = float(torch.quantile((got - ref64).abs().flatten()[:4_000_000], 0.9))
    return e_got <= max(5e-3, 3 * e_ref) and p90 <= 1e-4 * rms, (e_got, e_ref, p90 / max(rms, 1e-300))
