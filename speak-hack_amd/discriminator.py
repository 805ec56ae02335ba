"""``StyleDiscriminator`` / ``DiscriminatorBlock`` (styleganv1.py:637-695) -- SURVEY.md 8f row F2.

Same constructor, attribute names and ``state_dict`` layout as the reference (spectral-norm wrapped
convs / linears: ``*.weight_orig``, ``*.weight_u``, ``*.weight_v``, ``*.bias``).  Forward runs on the HIP
conv / FC kernels (3x3 s1 and 3x3 s2 with fused bias + LeakyReLU); the spectral normalisation of ALL wrapped layers
(one power iteration on each [Cout, Cin*k*k] matrix in training mode, then ``weight_orig / sigma`` -- what
torch.nn.utils.spectral_norm's pre-forward hook does with ~a dozen launches per layer) is one grouped call
(``autograd.SpectralNormAllFn``, csrc/spectral_norm.hip) whose backward carries parameter gradients to ``weight_orig``.  Backward runs on the HIP
epilogue-adjoint / dgrad / wgrad kernels (``autograd.ConvBiasLReLUFn``); the R1 penalty's double backward
(train.py:246-255: ``autograd.grad(D(x).sum(), x, create_graph=True)``) runs on ``autograd.ConvDgradFn``, whose
adjoints are again the forward-conv and wgrad kernels.
"""
from __future__ import annotations

import logging
import math

import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from . import autograd as A
from . import ops

LRELU = 0.2


def _sn_weight(m: nn.Module) -> torch.Tensor:
    """Run the module's spectral-norm pre-hook (power iteration in training mode, exactly as a call of
    the module would) and return the normalised weight it installs."""
    for hook in m._forward_pre_hooks.values():
        hook(m, None)
    return m.weight


def _conv_lrelu(conv: nn.Conv2d, x, lrelu=True, weight=None):
    """``weight``: the spectrally normalised weight when the caller computed it (all layers in one grouped call);
    None: this module's own pre-forward hook runs (stand-alone use of a block)."""
    if weight is None:
        weight = _sn_weight(conv).contiguous()
    return A.conv_bias_lrelu(x, weight, conv.bias, conv.kernel_size[0], conv.stride[0], LRELU if lrelu else None)


class DiscriminatorBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = spectral_norm(nn.Conv2d(in_channels, in_channels, kernel_size=3, padding=1))
        self.conv2 = spectral_norm(nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1, stride=2))

    def forward(self, x, weights=None):
        w1, w2 = weights if weights is not None else (None, None)
        return _conv_lrelu(self.conv2, _conv_lrelu(self.conv1, x, weight=w1), weight=w2)


class StyleDiscriminator(nn.Module):
    def __init__(self, resolution=256, fmap_base=8192, num_channels=3, fmap_max=512):
        super().__init__()
        self.resolution_log2 = int(math.log2(resolution))
        self.nf = lambda stage: min(int(fmap_base / (2.0 ** stage)), fmap_max)
        self.fromrgb = spectral_norm(nn.Conv2d(num_channels, self.nf(self.resolution_log2 - 1), kernel_size=1))
        self.blocks = nn.ModuleList(DiscriminatorBlock(self.nf(res - 1), self.nf(res - 2))
                                    for res in range(self.resolution_log2, 2, -1))
        self.final_conv = spectral_norm(nn.Conv2d(self.nf(2), self.nf(1), kernel_size=3, padding=1))
        self.adaptive_pool = nn.AdaptiveAvgPool2d((1, 1))
        self.dense0 = spectral_norm(nn.Linear(self.nf(1), self.nf(0)))
        self.dense1 = spectral_norm(nn.Linear(self.nf(0), 1))
        self.logger = logging.getLogger(__name__)

    def _wrapped(self):
        return [self.fromrgb] + [c for b in self.blocks for c in (b.conv1, b.conv2)] + [self.final_conv, self.dense0, self.dense1]

    def forward(self, x):
        # every wrapped layer's power iteration + W / sigma in one grouped call (5 launches instead of ~200 per pass)
        ws = A.spectral_norm_all(self._wrapped(), self.training)
        x = _conv_lrelu(self.fromrgb, x.contiguous(), weight=ws[0])
        for i, block in enumerate(self.blocks):
            x = block(x, (ws[1 + 2 * i], ws[2 + 2 * i]))
        x = _conv_lrelu(self.final_conv, x, weight=ws[-3])
        x = A.global_avgpool(x).view(x.size(0), -1)
        x = A.fc(x, ws[-2], self.dense0.bias, 1.0, 1.0, LRELU)
        return A.fc(x, ws[-1], self.dense1.bias, 1.0, 1.0, 1.0)
