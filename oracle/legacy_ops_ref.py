"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's standalone StyleGAN1 ops
(the only definitions it has of the north-star's PixelNorm / FIR-blur / upscale family).
PINNED by tests/golden/legacy_ops.npz.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def pixel_norm(x, epsilon: float = 1e-8):
    """``PixelNorm.forward`` -- styleganv1.py:132-136: x * rsqrt(mean_c(x^2) + eps)."""
    return x * torch.rsqrt(torch.mean(x * x, dim=1, keepdim=True) + epsilon)


def pixel_norm_sqrt(x, epsilon: float = 1e-8):
    """``stylegan.PixelNorm.forward`` -- stylegan.py:28-29: x / sqrt(mean_c(x^2) + eps)."""
    return x / torch.sqrt(torch.mean(x ** 2, dim=1, keepdim=True) + epsilon)


def instance_norm(x, epsilon: float = 1e-8):
    """``InstanceNorm.forward`` -- styleganv1.py:148-152 (biased variance, eps inside rsqrt)."""
    x = x - torch.mean(x, (2, 3), True)
    return x * torch.rsqrt(torch.mean(x * x, (2, 3), True) + epsilon)


def blur2d_kernel(f=(1, 2, 1), normalize=True, flip=False):
    """``Blur2d.__init__`` -- styleganv1.py:38-47: separable outer product, normalised."""
    f = torch.tensor(list(f), dtype=torch.float32)
    k = f[:, None] * f[None, :]
    if normalize:
        k = k / k.sum()
    if flip:
        k = torch.flip(k, [0, 1])
    return k


def blur2d(x, f=(1, 2, 1), normalize=True, flip=False, stride=1):
    """``Blur2d.forward`` -- styleganv1.py:52-63: depthwise FIR, zero padding (k-1)//2."""
    k = blur2d_kernel(f, normalize, flip)
    C = x.size(1)
    return F.conv2d(x, k[None, None].expand(C, -1, -1, -1), stride=stride,
                    padding=int((k.size(0) - 1) / 2), groups=C)


def upscale2d(x, factor=2, gain=1):
    """``Upscale2d.forward`` -- styleganv1.py:113-120: optional gain, nearest-neighbour repeat."""
    if gain != 1:
        x = x * gain
    if factor > 1:
        x = x.repeat_interleave(factor, dim=2).repeat_interleave(factor, dim=3)
    return x


def fused_upscale(x, weight, bias=None):
    """``GBlock.up_sample`` for res >= 7 -- styleganv1.py:231,258: ``nn.ConvTranspose2d(Cin, Cout, 4, stride=2,
    padding=1)``; ``weight`` is [Cin,Cout,4,4].  Restated as the scatter it is: every input pixel adds its 4x4 stamp at
    (2y-1, 2x-1)."""
    B, Cin, H, W = x.shape
    Cout = weight.shape[1]
    full = x.new_zeros(B, Cout, 2 * H + 2, 2 * W + 2)        # output rows -1 .. 2H: cropped below
    stamp = torch.einsum("bihw,iokl->bohwkl", x, weight)
    for ky in range(4):
        for kx in range(4):
            full[:, :, ky:ky + 2 * H:2, kx:kx + 2 * W:2] += stamp[..., ky, kx]
    y = full[:, :, 1:2 * H + 1, 1:2 * W + 1]
    return y + bias.view(1, -1, 1, 1) if bias is not None else y
