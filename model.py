"""Drop-in for the reference's ``model.py``: ``from model import IRFD, IRFDLoss, StyleGANLoss``
(train.py:12, inference.py:8, test_irfd.py:6) keeps working, with ``IRFD`` on the MI355X kernels.

``IRFD`` -- the hot path (encoders -> latent -> decoder) -- lives in ``speak-hack_amd/irfd.py``.
``StyleGANLoss`` is the reference's two-term MSE (model.py:130-137), plain tensor arithmetic.
``IRFDLoss`` (model.py:186-386) is outside the accelerated path and cannot be rebuilt offline: it
needs dlib + a downloaded landmark model, hsemotion_onnx, and 6DRepNet weights fetched by URL
(SURVEY.md 2 rows 7-8); constructing it raises with that explanation.  Benchmarks and tests use the
reconstruction term ``mean((x_recon - x)^2)`` it is built around.
"""
import importlib as _importlib

import torch
import torch.nn as nn

_irfd = _importlib.import_module("speak-hack_amd.irfd")
IRFD = _irfd.IRFD


class StyleGANLoss(nn.Module):
    def __init__(self, device):
        super().__init__()
        self.device = device
        self.criterion = nn.MSELoss()

    def forward(self, real, fake):
        return self.criterion(fake, torch.ones_like(fake)) + self.criterion(real, torch.zeros_like(real))


class IRFDLoss(nn.Module):
    def __init__(self, config=None, device=None):
        super().__init__()
        raise NotImplementedError(
            "IRFDLoss depends on dlib, hsemotion_onnx and network-fetched 6DRepNet / landmark weights and is "
            "outside the MI355X hot path (DESIGN.md 7). Use a reconstruction loss on IRFD's outputs instead.")


__all__ = ["IRFD", "IRFDLoss", "StyleGANLoss"]
