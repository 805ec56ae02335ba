"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the build-defined StyleGAN2 decoder variant
(SURVEY.md 8a A11 / 8f F1).  PARITY UNPINNED: the reference contains no StyleGAN2 code (only prose in
reference/styleganv2.txt:1835,1912); this follows the published formulas (Karras et al. 2020, and the
widely used stylegan2-pytorch module layout): weight modulation w' = s_i * w, demodulation
w'' = w' * rsqrt(sum w'^2 + 1e-8), upfirdn2d with the [1,3,3,1] FIR, noise injection, FusedLeakyReLU
(lrelu(x + b) * sqrt 2), skip-connection toRGB.  Channel schedule and I/O signature are those of the
reference's live decoder (styleganv1.py:575: nf(s) = min(8192 / 2^s, 512); [B,6144] -> [B,3,256,256]).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


def make_kernel(k=(1, 3, 3, 1)):
    k = torch.tensor(k, dtype=torch.float32)
    k = k[None, :] * k[:, None]
    return k / k.sum()


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """Zero-insert by ``up``, pad (negative = crop), correlate with the flipped kernel, decimate by ``down``."""
    B, C, H, W = x.shape
    if up > 1:
        z = x.new_zeros(B, C, H, up, W, up)
        z[:, :, :, 0, :, 0] = x
        x = z.view(B, C, H * up, W * up)
    p0, p1 = pad
    x = F.pad(x, (max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)))
    x = x[:, :, max(-p0, 0):x.shape[2] - max(-p1, 0), max(-p0, 0):x.shape[3] - max(-p1, 0)]
    w = torch.flip(kernel.to(x.dtype), [0, 1])[None, None].expand(C, -1, -1, -1)
    return F.conv2d(x, w, groups=C)[:, :, ::down, ::down]


def upsample2x(x, kernel1d=(1, 3, 3, 1)):
    """stylegan2 ``Upsample(blur_kernel)``: kernel * up^2, pad = ((k - up)//2 + up - 1, (k - up)//2)."""
    k = make_kernel(kernel1d) * 4
    p = k.shape[0] - 2
    return upfirdn2d(x, k, up=2, down=1, pad=((p + 1) // 2 + 1, p // 2))


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    scale = (1 / math.sqrt(weight.shape[1])) * lr_mul
    out = F.linear(x, weight * scale)
    if activation:
        return F.leaky_relu(out + bias * lr_mul, 0.2) * SQRT2
    return out + bias * lr_mul


def modulated_conv2d(x, weight, s, demodulate=True):
    """weight [Cout,Cin,k,k]; s [B,Cin] -- per-sample weights applied as one grouped convolution."""
    B, Cin, H, W = x.shape
    Cout, _, k, _ = weight.shape
    scale = 1 / math.sqrt(Cin * k * k)
    w = scale * weight[None] * s[:, None, :, None, None]
    if demodulate:
        w = w * torch.rsqrt(w.pow(2).sum([2, 3, 4], keepdim=True) + 1e-8)
    out = F.conv2d(x.reshape(1, B * Cin, H, W), w.reshape(B * Cout, Cin, k, k), padding=k // 2, groups=B)
    return out.view(B, Cout, H, W)


def styled_conv(x, w_lat, sd, p, noise, upsample):
    if upsample:
        x = upsample2x(x)
    s = equal_linear(w_lat, sd[p + "conv.modulation.weight"], sd[p + "conv.modulation.bias"])
    out = modulated_conv2d(x, sd[p + "conv.weight"], s, True)
    out = out + sd[p + "noise.weight"] * noise
    return F.leaky_relu(out + sd[p + "activate.bias"].view(1, -1, 1, 1), 0.2) * SQRT2


def to_rgb(x, w_lat, sd, p, skip):
    s = equal_linear(w_lat, sd[p + "conv.modulation.weight"], sd[p + "conv.modulation.bias"])
    out = modulated_conv2d(x, sd[p + "conv.weight"], s, False) + sd[p + "bias"].view(1, -1, 1, 1)
    if skip is not None:
        out = out + upsample2x(skip)
    return out


def channels(resolution=256, fmap_base=8192, fmap_max=512):
    """{res: channels} with the reference's schedule: 4x4 -> nf(1), res 2^r -> nf(r-1)."""
    nf = lambda stage: min(int(fmap_base / (2.0 ** stage)), fmap_max)
    return {2 ** r: nf(r - 1) for r in range(2, int(math.log2(resolution)) + 1)}


def generator(features, sd, noises, resolution=256, n_mlp=8):
    """[B,input_dim] -> [B,3,res,res].  ``noises``: 1 + 2*(log2(res)-2) tensors [B,1,r,r] in layer order."""
    x = features * torch.rsqrt(torch.mean(features ** 2, dim=1, keepdim=True) + 1e-8)          # PixelNorm
    for i in range(n_mlp):
        x = equal_linear(x, sd[f"style.{i}.weight"], sd[f"style.{i}.bias"], 0.01, True)
    w = x
    B = w.size(0)
    out = sd["input.input"].expand(B, -1, -1, -1)
    out = styled_conv(out, w, sd, "conv1.", noises[0], False)
    skip = to_rgb(out, w, sd, "to_rgb1.", None)
    for i in range(int(math.log2(resolution)) - 2):
        out = styled_conv(out, w, sd, f"convs.{2 * i}.", noises[1 + 2 * i], True)
        out = styled_conv(out, w, sd, f"convs.{2 * i + 1}.", noises[2 + 2 * i], False)
        skip = to_rgb(out, w, sd, f"to_rgbs.{i}.", skip)
    return skip


def noise_shapes(B, resolution=256):
    shapes = [(B, 1, 4, 4)]
    for r in range(3, int(math.log2(resolution)) + 1):
        shapes += [(B, 1, 2 ** r, 2 ** r)] * 2
    return shapes
