"""The stand-alone StyleGAN1 ops of the reference's legacy ``G_synthesis`` port (styleganv1.py:29-152):
``Blur2d``, ``Upscale2d``, ``PixelNorm``, ``InstanceNorm`` -- same constructors and forward signatures, on
the HIP kernels.  (The ``G_synthesis`` graph itself is dead code in the reference -- nothing instantiates it
and its constructor needs a CUDA device -- and is not rebuilt)"""
from __future__ import annotations

import torch
import torch.nn as nn

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
        return ops.blur2d(x.contiguous(), self.f[0, 0], self.stride)


class Upscale2d(nn.Module):
    def __init__(self, factor=2, gain=1):
        super().__init__()
        self.gain, self.factor = gain, factor

    def forward(self, x):
        if self.factor <= 1 and self.gain == 1:
            return x
        return ops.upscale2d_nearest(x.contiguous(), max(self.factor, 1), self.gain)


class PixelNorm(nn.Module):
    def __init__(self, epsilon=1e-8):
        super().__init__()
        self.epsilon = epsilon

    def forward(self, x):
        return ops.pixelnorm(x.contiguous(), self.epsilon, sqrt_form=False)


class InstanceNorm(nn.Module):
    def __init__(self, epsilon=1e-8):
        super().__init__()
        self.epsilon = epsilon

    def forward(self, x):
        return ops.instance_norm_affine(x.contiguous(), None, None, self.epsilon)
