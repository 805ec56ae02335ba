"""Mirror of the reference's ``stylegan.py`` (ProGAN-style StyleGAN1): the same
class names, constructor arguments, ``forward`` signatures and ``state_dict`` keys (145 entries for
``Generator(512, 512)``, ``rgb_layers.0`` aliasing ``initial_rgb`` as in the reference).  Forward and backward
on the HIP kernels (backward through ``autograd.FusedConvFn`` / ``InstanceNormAffineFn`` / ``ToRGBFn`` / ``FCFn``):

* ``WSConv2d`` = the MFMA conv with ``out_scale`` folding the ``x * scale`` pre-multiply (stylegan.py:45-46);
  inside ``GenBlock`` the bilinear x2, bias, noise and LeakyReLU ride in the same launch;
* ``AdaIN`` = one per-plane kernel (instance-norm statistics + style scale/shift, stylegan.py:91-95);
* rgb heads are the small-Cout 1x1 kernel; the fade-in is one tanh blend.  ``rgb(upsample(x))`` is computed as
  ``upsample(rgb(x))`` (a 1x1 conv and a bilinear resize commute) -- 4x fewer pixels through the 1x1.
The reference ``Discriminator`` (stylegan.py:181-263) is out of scope; the class is kept as
a parameter holder so checkpoints and imports resolve.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import autograd as AG
from . import ops

factors = [1, 1, 1, 1, 1 / 2, 1 / 4, 1 / 8, 1 / 16, 1 / 32]


class WSLinear(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features)
        self.scale = (2 / in_features) ** 0.5
        self.bias = self.linear.bias
        self.linear.bias = None
        nn.init.normal_(self.linear.weight)
        nn.init.zeros_(self.bias)

    def forward(self, x, relu=False):
        return AG.fc(x.contiguous(), self.linear.weight, self.bias, self.scale, 1.0, 0.0 if relu else 1.0)


class PixelNorm(nn.Module):
    def __init__(self):
        super().__init__()
        self.epsilon = 1e-8

    def forward(self, x):
        return AG.pixelnorm(x.contiguous(), self.epsilon, True)       # forward and adjoint on the HIP kernels


class WSConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding)
        self.scale = (2 / (in_channels * (kernel_size ** 2))) ** 0.5
        self.bias = self.conv.bias
        self.conv.bias = None
        nn.init.normal_(self.conv.weight)
        nn.init.zeros_(self.bias)
        self._pk = ops.PackedConvWeight()

    def forward(self, x, noise_w=None, noise=None, lrelu=None, upsample=False):
        w = self.conv.weight
        Cout, Cin, k, _ = w.shape
        train = torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)
        if k == 1 and Cout <= 4:
            if train:       # conv(x * scale, w) = conv(x, w * scale): the scale rides on the (tiny) weight tensor
                return AG.to_rgb(x.contiguous(), w * self.scale, self.bias)
            return ops.conv1x1_small(x.contiguous(), w, self.bias, in_scale=self.scale)
        if self.conv.padding[0] != (k - 1) // 2 or k not in (1, 3):
            raise NotImplementedError(f"WSConv2d: kernel {k} / padding {self.conv.padding} is not on the HIP path")
        if train:
            if k != 3:
                raise NotImplementedError("WSConv2d backward: 3x3 and the 1x1 toRGB are on the HIP path")
            # the cache is keyed on the Parameter itself; the equalised-lr scale rides on the accumulator (and on dx / dw)
            return AG.fused_conv(x.contiguous(), w, self.bias, noise_w, noise, None, upsample, lrelu, self._pk, w_scale=self.scale)
        B, _, H, W = x.shape
        Ho, Wo = (2 * H, 2 * W) if upsample else (H, W)
        cfg = ops.conv2d_pick_config(k, 1, B, Cin, Cout, Ho, Wo)
        return ops.conv2d_fused(x.contiguous(), self._pk.get(w, cfg), Cout, k, 1, bias=self.bias, noise_w=noise_w, noise=noise,
                                lrelu_slope=lrelu, out_scale=self.scale, upsample=upsample, config=cfg)


class MappingNetwork(nn.Module):
    def __init__(self, z_dim, w_dim):
        super().__init__()
        layers = [PixelNorm(), WSLinear(z_dim, w_dim)]
        for _ in range(7):
            layers += [nn.ReLU(), WSLinear(w_dim, w_dim)]
        self.mapping = nn.Sequential(*layers)

    def forward(self, x):
        mods = list(self.mapping)
        x = mods[0](x)
        i = 1
        while i < len(mods):                       # WSLinear followed by ReLU: one launch (slope 0)
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = mods[i](x, relu=fuse)
            i += 2 if fuse else 1
        return x


class InjectNoise(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1, channels, 1, 1))

    def forward(self, x, zero_noise=False, noise=None):
        if zero_noise:
            return x
        if noise is None:
            noise = torch.randn((x.shape[0], 1, x.shape[2], x.shape[3]), device=x.device)
        return AG.bias_noise_style(x.contiguous(), None, self.weight.view(-1), noise.contiguous(), None, noise.shape[0])


class AdaIN(nn.Module):
    def __init__(self, channels, w_dim):
        super().__init__()
        self.instance_norm = nn.InstanceNorm2d(channels)
        self.style_scale = WSLinear(w_dim, channels)
        self.style_bias = WSLinear(w_dim, channels)

    def forward(self, x, w):
        if x.size(0) != w.size(0):                 # the 4x4 constant is [1,C,4,4]: normalise once, style per sample
            x = x.expand(w.size(0), -1, -1, -1).contiguous()
        return AG.instance_norm_affine(x.contiguous(), self.style_scale(w), self.style_bias(w), self.instance_norm.eps)


class GenBlock(nn.Module):
    def __init__(self, in_channels, out_channels, w_dim):
        super().__init__()
        self.conv1 = WSConv2d(in_channels, out_channels)
        self.conv2 = WSConv2d(out_channels, out_channels)
        self.leaky = nn.LeakyReLU(0.2, inplace=True)
        self.inject_noise1 = InjectNoise(out_channels)
        self.inject_noise2 = InjectNoise(out_channels)
        self.adain1 = AdaIN(out_channels, w_dim)
        self.adain2 = AdaIN(out_channels, w_dim)

    def _conv(self, conv, inj, x, zero_noise, noise, upsample):
        if zero_noise:
            return conv(x, lrelu=0.2, upsample=upsample)
        B, H, W = x.shape[0], x.shape[2] * (2 if upsample else 1), x.shape[3] * (2 if upsample else 1)
        if noise is None:
            noise = torch.randn((B, 1, H, W), device=x.device)
        return conv(x, noise_w=inj.weight.view(-1), noise=noise.contiguous(), lrelu=0.2, upsample=upsample)

    def forward(self, x, w, zero_noise=False, noises=(None, None), upsample=False):
        """``upsample``: x is the block's input BEFORE the reference's F.interpolate (folded into conv1)."""
        x = self.adain1(self._conv(self.conv1, self.inject_noise1, x, zero_noise, noises[0], upsample), w)
        return self.adain2(self._conv(self.conv2, self.inject_noise2, x, zero_noise, noises[1], False), w)


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = WSConv2d(in_channels, out_channels)
        self.conv2 = WSConv2d(out_channels, out_channels)
        self.leaky = nn.LeakyReLU(0.2)

    def forward(self, x):
        return self.conv2(self.conv1(x, lrelu=0.2), lrelu=0.2)


class Generator(nn.Module):
    def __init__(self, w_dim, in_channels, img_channels=3):
        super().__init__()
        self.starting_constant = nn.Parameter(torch.ones((1, in_channels, 4, 4)))
        self.initial_adain1 = AdaIN(in_channels, w_dim)
        self.initial_adain2 = AdaIN(in_channels, w_dim)
        self.initial_noise1 = InjectNoise(in_channels)
        self.initial_noise2 = InjectNoise(in_channels)
        self.initial_conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.leaky = nn.LeakyReLU(0.2, inplace=True)
        self.initial_rgb = WSConv2d(in_channels, img_channels, kernel_size=1, stride=1, padding=0)
        self.prog_blocks, self.rgb_layers = nn.ModuleList([]), nn.ModuleList([self.initial_rgb])
        for i in range(len(factors) - 1):
            conv_in_c, conv_out_c = int(in_channels * factors[i]), int(in_channels * factors[i + 1])
            self.prog_blocks.append(GenBlock(conv_in_c, conv_out_c, w_dim))
            self.rgb_layers.append(WSConv2d(conv_out_c, img_channels, kernel_size=1, stride=1, padding=0))
        self._pk0 = ops.PackedConvWeight()

    def fade_in(self, alpha, upscaled, generated):
        if torch.is_grad_enabled() and (upscaled.requires_grad or generated.requires_grad):
            return torch.tanh(alpha * generated + (1 - alpha) * upscaled)       # [B,3,H,W] images: torch elementwise
        return ops.fade_in_tanh(generated.contiguous(), upscaled.contiguous(), alpha)

    def forward(self, w, alpha, steps, zero_noise=False, noises=None):
        """``noises`` (optional): the 2 + 2*steps noise tensors in the reference's draw order (stylegan.py:81)."""
        nz = iter(noises) if noises is not None else None
        nxt = (lambda: next(nz)) if nz is not None else (lambda: None)
        B = w.size(0)
        x = self.initial_adain1(self.initial_noise1(self.starting_constant, zero_noise, None if zero_noise else nxt()), w)
        Cc = x.size(1)
        n2 = None if zero_noise else nxt()
        if not zero_noise and n2 is None:
            n2 = torch.randn((B, 1, 4, 4), device=w.device)
        # initial_conv (plain nn.Conv2d, stylegan.py:135) -> x; leaky(noise2(x)) -> adain2 -> out (stylegan.py:161-162).
        # Reference quirk kept: the LeakyReLU is in-place (stylegan.py:136) and with zero_noise InjectNoise returns
        # its input, so `x` itself is overwritten before `initial_rgb(x)` at steps == 0; with noise it is not.
        cw, cb = self.initial_conv.weight, self.initial_conv.bias

        def conv0(noise_w, noise, slope):
            return AG.fused_conv(x.contiguous(), cw, cb, noise_w, noise, None, False, slope, self._pk0)

        if zero_noise:
            pre = conv0(None, None, 0.2)
            if steps == 0:
                return self.initial_rgb(pre)
        else:
            if steps == 0:
                return self.initial_rgb(conv0(None, None, None))
            pre = conv0(self.initial_noise2.weight.view(-1), n2.contiguous(), 0.2)
        out = self.initial_adain2(pre, w)
        prev = out
        for step in range(steps):
            prev = out                               # the tensor the reference upsamples at this step
            ns = (None, None) if zero_noise else (nxt(), nxt())
            out = self.prog_blocks[step](out, w, zero_noise, ns, upsample=True)
        # rgb(upsample(prev)) == upsample(rgb(prev)): a 1x1 conv commutes with the bilinear resize
        final_upscaled = AG.upsample2x(self.rgb_layers[steps - 1](prev))
        final_out = self.rgb_layers[steps](out)
        return self.fade_in(alpha, final_upscaled, final_out)


class Discriminator(nn.Module):
    """Parameter holder with the reference's layout (stylegan.py:181-218); forward is out of scope."""

    def __init__(self, in_channels, img_channels=3):
        super().__init__()
        self.prog_blocks, self.rgb_layers = nn.ModuleList([]), nn.ModuleList([])
        self.leaky = nn.LeakyReLU(0.2)
        for i in range(len(factors) - 1, 0, -1):
            conv_in, conv_out = int(in_channels * factors[i]), int(in_channels * factors[i - 1])
            self.prog_blocks.append(ConvBlock(conv_in, conv_out))
            self.rgb_layers.append(WSConv2d(img_channels, conv_in, kernel_size=1, stride=1, padding=0))
        self.initial_rgb = WSConv2d(img_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.rgb_layers.append(self.initial_rgb)
        self.avg_pool = nn.AvgPool2d(kernel_size=2, stride=2)
        self.final_block = nn.Sequential(WSConv2d(in_channels + 1, in_channels, kernel_size=3, padding=1), nn.LeakyReLU(0.2),
                                         WSConv2d(in_channels, in_channels, kernel_size=4, padding=0, stride=1),
                                         nn.LeakyReLU(0.2), WSConv2d(in_channels, 1, kernel_size=1, padding=0, stride=1))

    def forward(self, x, alpha, steps):
        raise NotImplementedError("stylegan.Discriminator is outside the accelerated path (SURVEY.md 2 row 4)")
