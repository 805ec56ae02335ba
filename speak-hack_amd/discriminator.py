"""``StyleDiscriminator`` / ``DiscriminatorBlock`` (styleganv1.py:637-695) -- SURVEY.md 8f row F2.

Same constructor, attribute names and ``state_dict`` layout as the reference (spectral-norm wrapped
convs / linears: ``*.weight_orig``, ``*.weight_u``, ``*.weight_v``, ``*.bias``).  Forward runs on the HIP
conv / FC kernels (3x3 s1 and 3x3 s2 with fused bias + LeakyReLU); the spectral normalisation itself
(one power iteration on a [Cout, Cin*k*k] matrix in training mode, torch.nn.utils.spectral_norm's
own hook) is a few tiny matrix-vector products and stays on torch.  Backward (incl. the R1
double-backward of train.py:246-255) is not built yet.
"""
from __future__ import annotations

import logging
import math

import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from . import ops

LRELU = 0.2


def _sn_weight(m: nn.Module) -> torch.Tensor:
    """Run the module's spectral-norm pre-hook (power iteration in training mode, exactly as a call of
    the module would) and return the normalised weight it installs."""
    for hook in m._forward_pre_hooks.values():
        hook(m, None)
    return m.weight


def _conv_lrelu(conv: nn.Conv2d, x, lrelu=True):
    w = _sn_weight(conv).detach().contiguous()
    k, stride = conv.kernel_size[0], conv.stride[0]
    B, Cin, H, W = x.shape
    Cout = conv.out_channels
    Ho, Wo = ops.conv_out_size(H, k, stride), ops.conv_out_size(W, k, stride)
    cfg = ops.conv2d_pick_config(k, stride, B, Cin, Cout, Ho, Wo)
    return ops.conv2d_fused(x, ops.pack_conv_weight(w, cfg), Cout, k, stride, bias=conv.bias,
                            lrelu_slope=LRELU if lrelu else None, config=cfg)


class DiscriminatorBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = spectral_norm(nn.Conv2d(in_channels, in_channels, kernel_size=3, padding=1))
        self.conv2 = spectral_norm(nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1, stride=2))

    def forward(self, x):
        return _conv_lrelu(self.conv2, _conv_lrelu(self.conv1, x))


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

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("StyleDiscriminator backward is not built yet (SURVEY.md 8f F2); "
                                      "call it under torch.no_grad()")
        x = _conv_lrelu(self.fromrgb, x.contiguous())
        for block in self.blocks:
            x = block(x)
        x = _conv_lrelu(self.final_conv, x)
        x = ops.global_avgpool(x).view(x.size(0), -1)
        x = ops.fc(x, _sn_weight(self.dense0).detach().contiguous(), self.dense0.bias, slope=LRELU)
        return ops.fc(x, _sn_weight(self.dense1).detach().contiguous(), self.dense1.bias)
