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
