"""The stand-alone StyleGAN1 ops of the reference's legacy ``G_synthesis`` port (styleganv1.py:29-152):
``Blur2d``, ``Upscale2d``, ``PixelNorm``, ``InstanceNorm`` -- same constructors and forward signatures, on
the HIP kernels, forward and backward (``autograd.Blur2dFn`` / ``Upscale2dFn`` / ``PixelNormFn`` /
``InstanceNormAffineFn``) -- and ``FusedUpscale``, the ``nn.ConvTranspose2d(4, stride=2, padding=1)`` that
``GBlock`` uses as its upsampler from 128^2 on (styleganv1.py:231,258), forward on the MFMA parity kernels, backward on
the 4x4 stride-2 conv / weight-gradient kernels (``autograd.FusedUpscaleFn``).
(The ``G_synthesis`` graph itself is dead code in the reference -- nothing instantiates it
and its constructor needs a CUDA device -- and is not rebuilt)"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import autograd as AG
from . import ops


class Blur2d(nn.Module):
    def __init__(self, f=[1, 2, 1], normalize=True, flip=False, stride=1):
        super().__init__()
        assert isinstance(f, list) or f is None, "kernel f must be an instance of python built_in type list!"
        if f is not None:
            f = torch.tensor(f, dtype=torch.float32)
            f = f[:, None] * f[None, :]
            if normalize:
                f = f / f.sum()
            if flip:
                f = torch.flip(f, [0, 1])
            self.f = f[None, None]
        else:
            self.f = None
        self.stride = stride

    def forward(self, x):
        if self.f is None:
            return x
        return AG.blur2d(x.contiguous(), self.f[0, 0], self.stride)


class Upscale2d(nn.Module):
    def __init__(self, factor=2, gain=1):
        super().__init__()
        self.gain, self.factor = gain, factor

    def forward(self, x):
        if self.factor <= 1 and self.gain == 1:
            return x
        return AG.upscale2d(x.contiguous(), max(self.factor, 1), self.gain)


class PixelNorm(nn.Module):
    def __init__(self, epsilon=1e-8):
        super().__init__()
        self.epsilon = epsilon

    def forward(self, x):
        return AG.pixelnorm(x.contiguous(), self.epsilon, False)


class InstanceNorm(nn.Module):
    def __init__(self, epsilon=1e-8):
        super().__init__()
        self.epsilon = epsilon

    def forward(self, x):
        return AG.instance_norm_affine(x.contiguous(), None, None, self.epsilon)


class FusedUpscale(nn.ConvTranspose2d):
    """``GBlock.up_sample`` for res >= 7 (styleganv1.py:231: ``nn.ConvTranspose2d(nf(res-3), nf(res-2), 4, stride=2,
    padding=1)``): same parameters and ``state_dict`` keys (``weight`` [Cin,Cout,4,4], ``bias``); the forward runs as four
    output-parity 2x2 MFMA kernels in one launch (include/spk.h, SPK_CONV_TRANSPOSE4X4_S2); the backward
    (``autograd.FusedUpscaleFn``) on the 4x4 stride-2 conv / weight-gradient kernels."""

    def __init__(self, in_channels, out_channels):
        super().__init__(in_channels, out_channels, 4, stride=2, padding=1)
        self._pk = ops.PackedConvWeight()

    def forward(self, x):
        return AG.fused_upscale(x.contiguous(), self.weight, self.bias, self._pk)
