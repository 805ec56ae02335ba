import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The upsample-folded weight gradient (csrc/wgrad_mfma_f32.hip, wgrad3x3_up_kernel) is by default used for planes at least 128
# wide, where it pays (spk_conv2d_wgrad_up_supported; narrower upsample layers materialise the x2 image and run the plain
# kernel); the test process sends EVERY shape it can take through it (the library reads this once, at first use), so the
# decoder's block / end-to-end gradient tests cover it at 16^2..64^2 as well.  The SHIPPED dispatch for those widths is run by
# tests/test_backward_gpu.py::test_decoder_gradients_with_the_default_upsample_threshold in a child process.
os.environ.setdefault("SPK_WGRAD_UP_MIN_W", "16")


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


def grad_close(got, ref32, ref64, pixels=None, noise_floor=0.0, err_floor=0.0):
    """Mask-flip-aware gradient criterion for deep ReLU / LeakyReLU nets; both parts are against the fp64 evaluation of
    the reference.

    Where the noise comes from: the HIP path and the reference sum in different orders, so a pre-activation within rounding
    (~1e-6 of its scale) of zero can land on the other side and flip its mask.  With N activations about 1e-6 * N of them
    are that close: one or two per pass of the nets tested here, and whether the reference's OWN fp32 evaluation flips one
    is luck (``ref32`` is a single sample of that lottery).  What one flip does: at a layer with P = batch x pixels
    positions it removes / adds one position's contribution, i.e. it moves a DENSE gradient (a weight, bias or style
    gradient sums over all P positions -- and every second-order gradient couples all of them) by about 1/P of its size,
    everywhere at once; only activation-shaped gradients see a localised change.

      (1) rel-L2 <= max(5e-3, 3 * err(ref32)): a wrong slope, stride, tap order or a missing term is off by > 1e-1;
      (2) the 90th percentile of |difference| <= rms(reference) * max(1e-4, 4 * p90(ref32 error) / rms, 8 / P): outside
          the flipped receptive fields the agreement is at fp32 rounding level, and a dense gradient may carry the 1/P
          footprint of a few flips (``pixels`` = P of the largest layer, i.e. batch x input pixels; None = unknown: no
          flip allowance, the round-1 form).  The fixed 1e-4 of round 1 sat AT that footprint for the 32^2..64^2 test nets
          (1 / (2 * 64 * 64) = 1.2e-4) and tripped on any change of summation order.

    ``noise_floor`` (optional): a p90 / rms level established over a whole sub-network (see ``grad_stats`` and
    tests/test_irfd_gpu.py) that is allowed 4x like the parameter's own fp32 error.  ``err_floor`` (optional): a rel-L2 level
    of the reference's own fp32 evaluation established over the parameters of the same KIND (same test), standing in for
    ``err(ref32)`` in (1) where it is larger.

    Returns (ok, (err, err_ref32, p90/rms))."""
    e_got, e_ref, r_got, r_ref = grad_stats(got, ref32, ref64)
    bound = max(1e-4, 4 * r_ref, 4 * noise_floor, (8.0 / pixels) if pixels else 0.0)
    return e_got <= max(5e-3, 3 * max(e_ref, err_floor)) and r_got <= bound, (e_got, e_ref, r_got)


def grad_stats(got, ref32, ref64):
    """(rel-L2 of got, rel-L2 of ref32, p90|got - ref64| / rms, p90|ref32 - ref64| / rms), all against ``ref64``."""
    import torch
    got, ref32, ref64 = (t.detach().cpu().double() for t in (got, ref32, ref64))
    rms = max(float(ref64.pow(2).mean().sqrt()), 1e-300)
    p90 = float(torch.quantile((got - ref64).abs().flatten()[:4_000_000], 0.9))
    p90_ref = float(torch.quantile((ref32 - ref64).abs().flatten()[:4_000_000], 0.9))
    return rel_l2(got, ref64), rel_l2(ref32, ref64), p90 / rms, p90_ref / rms
