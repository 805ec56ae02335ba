"""Build-defined StyleGAN2 decoder variant -- the north-star's named kernels
(modulated 3x3 conv + demodulation, upfirdn2d, noise injection, FusedLeakyReLU, PixelNorm) on the HIP path.
The reference has no StyleGAN2 code; module layout and formulas follow the published
StyleGAN2 / the common stylegan2-pytorch naming, with the reference decoder's channel schedule and I/O
signature ([B,6144] latent -> [B,3,256,256]) so it can stand in for ``IRFD.Gd``.  Forward and backward
(``autograd.ModConvFn`` / ``ModToRGBFn`` / ``UpFirDnFn``): the data gradient of a modulated conv is the same modulated MFMA
conv with the roles of s and d swapped, the weight gradient the MFMA wgrad kernel with both factors applied while staging, the
demodulation and its adjoint small kernels of their own -- no rescaled activation and no ATen GEMM in a training step.

How the modulated conv maps to the MI355X kernel: StyleGAN2's per-sample weight
``w'' = w * s[b,ci] * d[b,co]`` is never formed.  ``s`` multiplies the *input* while it is staged into LDS
(SPK_CONV_IN_BATCH_SCALE), the demodulation ``d = rsqrt(sum (w*s)^2 + eps)`` (a tiny [B,Cout] kernel)
multiplies the *output* in the epilogue together with noise, bias, LeakyReLU and the sqrt(2) gain -- so the
packed weights are shared by the whole batch and the conv is the same implicit GEMM as everywhere else.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import autograd as AG
from . import ops

SQRT2 = math.sqrt(2.0)


def make_kernel(k=(1, 3, 3, 1)):
    k = torch.tensor(k, dtype=torch.float32)
    k = k[None, :] * k[:, None]
    return k / k.sum()


class EqualLinear(nn.Module):
    def __init__(self, in_dim, out_dim, bias_init=0.0, lr_mul=1.0, activation=False):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init))
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul, self.activation = lr_mul, activation

    def forward(self, x):
        if self.activation:     # sqrt2 * lrelu(v) == lrelu(sqrt2 * v): fold the gain into the FC's multipliers
            return AG.fc(x.contiguous(), self.weight, self.bias, self.scale * SQRT2, self.lr_mul * SQRT2, 0.2)
        return AG.fc(x.contiguous(), self.weight, self.bias, self.scale, self.lr_mul, 1.0)


class Upsample(nn.Module):
    def __init__(self, kernel=(1, 3, 3, 1), factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, x):
        if not hasattr(self, "_k_host"):
            self._k_host = self.kernel.detach().cpu()
        return AG.upfirdn(x.contiguous(), self._k_host, self.factor, 1, self.pad)


class ModulatedConv2d(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True):
        super().__init__()
        self.in_channel, self.out_channel, self.kernel_size, self.demodulate = in_channel, out_channel, kernel_size, demodulate
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1.0)
        self._pk = ops.PackedConvWeight()

    def forward(self, x, style, bias=None, noise_w=None, noise=None, lrelu=None, act_gain=1.0, upsample=False, s=None, d=None,
                skip=None, weight=None):
        """``s`` / ``d`` (optional): the modulation ``self.modulation(style)`` and the demodulation vector computed
        elsewhere (grouped launches).  ``skip`` (toRGB only): the previous resolution's image; its [1,3,3,1] x2 upsample
        and the sum happen inside the toRGB launch."""
        if s is None:
            s = self.modulation(style)
        if self.kernel_size == 1 and self.out_channel <= 4 and not self.demodulate:
            return AG.mod_to_rgb(x.contiguous(), self.weight, s.contiguous(), bias, self.scale,
                                 None if skip is None else skip.contiguous(), _UP_FIR)
        if self.kernel_size != 3:
            raise NotImplementedError("ModulatedConv2d: 3x3 (styled convs) and 1x1 toRGB are on the HIP path")
        if d is not None:        # a demodulation vector computed elsewhere (the grouped launch of the inference path)
            B, Cin, Hs, Ws = x.shape
            H, W = (2 * Hs, 2 * Ws) if upsample else (Hs, Ws)
            if ops.use_wino(B, Cin, self.out_channel, H, W) and (not upsample or Ws % 4 == 0):
                xin = ops.upsample2x(x, zero_border=True) if upsample else x.contiguous()
                return ops.conv3x3_wino(xin, self._pk.get_wino(self.weight), self.out_channel, bias=bias, noise_w=noise_w, noise=noise,
                                        lrelu_slope=lrelu, out_scale=self.scale, batch_scale=s.contiguous(), demod=d, act_gain=act_gain)
            cfg = ops.conv2d_pick_config(3, 1, B, Cin, self.out_channel, H, W)
            cfg = cfg + 4 if cfg < 4 else cfg
            return ops.conv2d_fused(x.contiguous(), self._pk.get(self.weight, cfg), self.out_channel, 3, 1, bias=bias, noise_w=noise_w,
                                    noise=noise, lrelu_slope=lrelu, out_scale=self.scale, batch_scale=s.contiguous(), demod=d,
                                    act_gain=act_gain, config=cfg, upsample=upsample, up_fir=True)
        # upfirdn2d(up=2, [1,3,3,1]) is folded into the conv's input staging: no 4x tensor in HBM; the demodulation vector and
        # its adjoint are kernels inside the Function (spk_modconv_demod / spk_modconv_demod_bwd)
        # (``weight``: this conv's weight behind the generator's AG.gate_weights node, a training pass)
        return AG.mod_conv(x.contiguous(), self.weight if weight is None else weight, s.contiguous(), bias, noise_w, noise, self.scale,
                           upsample, lrelu, act_gain, _UP_FIR, self._pk, demodulate=self.demodulate)


_UP_FIR = make_kernel((1, 3, 3, 1)) * 4.0        # host copy of the x2 FIR (gain up^2), for the backward's materialised passes


class NoiseInjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))


class StyledConv(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False):
        super().__init__()
        self.upsample = upsample
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, x, style, noise=None, s=None, d=None, weight=None):
        B, _, H, W = x.shape
        if self.upsample:
            H, W = 2 * H, 2 * W
        if noise is None:
            noise = torch.randn(B, 1, H, W, device=x.device)
        nw = self.noise.weight.expand(self.conv.out_channel).contiguous()
        # conv*demod + noise_w*noise + bias -> lrelu(0.2) * sqrt2: one launch
        return self.conv(x, style, bias=self.activate.bias, noise_w=nw, noise=noise.contiguous(), lrelu=0.2, act_gain=SQRT2,
                         upsample=self.upsample, s=s, d=d, weight=weight)


class ToRGB(nn.Module):
    def __init__(self, in_channel, style_dim, upsample=True):
        super().__init__()
        self.upsample = Upsample() if upsample else None
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, x, style, skip=None, s=None):
        if skip is not None and (self.upsample is None or skip.shape[-1] * 2 != x.shape[-1]):
            return self.conv(x, style, bias=self.bias.view(-1), s=s) + (skip if self.upsample is None else self.upsample(skip))
        # one launch: modulated 1x1 + bias + upfirdn2d(skip, up=2, [1,3,3,1]) + add
        return self.conv(x, style, bias=self.bias.view(-1), s=s, skip=skip)


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))


class StyleGAN2Generator(nn.Module):
    """[B,input_dim] -> [B,3,resolution,resolution]; same constructor-level knobs as ``StyleGenerator``."""

    def __init__(self, input_dim=6144, style_dim=512, n_mlp=8, resolution=256, fmap_base=8192, fmap_max=512):
        super().__init__()
        self.input_dim, self.style_dim, self.resolution = input_dim, style_dim, resolution
        nf = lambda stage: min(int(fmap_base / (2.0 ** stage)), fmap_max)
        self.style = nn.ModuleList(EqualLinear(input_dim if i == 0 else style_dim, style_dim, lr_mul=0.01, activation=True)
                                   for i in range(n_mlp))
        self.input = ConstantInput(nf(1))
        self.conv1 = StyledConv(nf(1), nf(1), 3, style_dim)
        self.to_rgb1 = ToRGB(nf(1), style_dim, upsample=False)
        self.convs, self.to_rgbs = nn.ModuleList(), nn.ModuleList()
        cin = nf(1)
        for r in range(3, int(math.log2(resolution)) + 1):
            cout = nf(r - 1)
            self.convs.append(StyledConv(cin, cout, 3, style_dim, upsample=True))
            self.convs.append(StyledConv(cout, cout, 3, style_dim))
            self.to_rgbs.append(ToRGB(cout, style_dim))
            cin = cout

    use_plan = True      # inference forwards go out as one pre-built launch list (plan.StyleGAN2Plan)
    precision = "f32"    # "bf16x3": opt-in split-precision convs for inference (see decoder.SynthesisNetwork.precision)

    def forward(self, features, noises=None):
        train = torch.is_grad_enabled() and (features.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not train and self.use_plan and features.is_cuda and features.dim() == 2:
            from . import plan as PL
            B = features.size(0)
            key = (B, features.device, torch.cuda.current_stream(features.device).cuda_stream, "features", self.precision)
            p = PL.plan_for(self, key, lambda: PL.StyleGAN2Plan(self, B, features.device, precision=self.precision))
            return p.run(features.contiguous(), None if noises is None else [n.contiguous() for n in noises])
        w = AG.pixelnorm(features.contiguous(), 1e-8, False)
        for layer in self.style:
            w = layer(w)
        B = w.size(0)
        nz = iter(noises) if noises is not None else None
        nxt = (lambda: next(nz)) if nz is not None else (lambda: None)
        # every modulation is an affine of the same w: all of them in two grouped launches instead of 20
        layers = [self.conv1, self.to_rgb1] + [m for i in range(len(self.to_rgbs))
                                               for m in (self.convs[2 * i], self.convs[2 * i + 1], self.to_rgbs[i])]
        mods = [m.conv.modulation for m in layers]
        if train:
            # one autograd node per <= 16 affines: grouped forward launch, two grouped backward launches (20 FCFn nodes cost
            # 20 + 40 launches and 20 accumulations into d w)
            ss = []
            for k in range(0, len(mods), ops.L.FC_MAX_GROUPS):
                part = mods[k:k + ops.L.FC_MAX_GROUPS]
                confs = tuple((float(m.scale), float(m.lr_mul), 1.0, True) for m in part)
                w3 = w.unsqueeze(1).expand(B, len(part), w.shape[1])
                ss += list(AG.StyleFCGroupFn.apply(w3, confs, torch.is_grad_enabled(), *[t for m in part for t in (m.weight, m.bias)]))
        else:
            ss = []
            for k in range(0, len(mods), ops.L.FC_MAX_GROUPS):
                ss += ops.fc_grouped((w, m.weight, m.bias, m.scale, m.lr_mul, 1.0) for m in mods[k:k + ops.L.FC_MAX_GROUPS])
        # ... and the demodulation vectors of the 13 styled convs only on those modulations: one more grouped launch
        styled = [m for m in layers if isinstance(m, StyledConv)]
        dd = [None] * len(styled)
        if not train and len(styled) <= ops.L.DEMOD_MAX_GROUPS:
            s_of = {id(m): sv for m, sv in zip(layers, ss)}
            dd = ops.modconv_demod_grouped((m.conv.weight, s_of[id(m)], m.conv.scale) for m in styled)
        sit, dit = iter(ss), iter(dd)
        # a training pass: the styled convs' weights go through ONE gate node, whose backward (after every conv's) joins the second
        # stream that their weight gradients are queued on (autograd.WeightGateFn)
        git = iter([None] * len(styled))
        if train and features.is_cuda and ops.side_stream(features.device) is not None and all(m.conv.weight.requires_grad for m in styled):
            git = iter(AG.gate_weights([m.conv.weight for m in styled]))
        out = self.input.input.expand(B, -1, -1, -1).contiguous()
        out = self.conv1(out, w, nxt(), s=next(sit), d=next(dit), weight=next(git))
        skip = self.to_rgb1(out, w, s=next(sit))
        for i, rgb in enumerate(self.to_rgbs):
            out = self.convs[2 * i](out, w, nxt(), s=next(sit), d=next(dit), weight=next(git))
            out = self.convs[2 * i + 1](out, w, nxt(), s=next(sit), d=next(dit), weight=next(git))
            skip = rgb(out, w, skip, s=next(sit))
        return skip
