"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's live decoder
(``styleganv1.StyleGenerator`` and everything it calls), as plain functions over a
``state_dict``.  fp32, NCHW.  PINNED by tests/golden/decoder_*.npz (tools/make_goldens.py).

Every function cites the reference lines it follows (paths relative to /root/reference).
Noise is always explicit: the reference draws ``torch.randn`` inside ``ApplyNoise.forward``
when ``noise is None`` (styleganv1.py:454-455); the golden generator records those draws.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.2


def wscale_fc(in_features: int, gain: float, use_wscale: bool, lrmul: float):
    """(w_lrmul, b_lrmul) runtime multipliers of ``FC`` -- styleganv1.py:474-485."""
    he_std = gain * in_features ** (-0.5)
    w_lrmul = he_std * lrmul if use_wscale else lrmul
    return w_lrmul, lrmul


def fc(x, weight, bias, w_lrmul: float, b_lrmul: float):
    """``FC.forward`` -- styleganv1.py:489-495: linear with runtime-scaled weight/bias, then
    LeakyReLU(0.2) (always, including for the style affine)."""
    out = F.linear(x, weight * w_lrmul, None if bias is None else bias * b_lrmul)
    return F.leaky_relu(out, LRELU_SLOPE)


def mapping(features, sd, prefix="mapping.", layers: int = 8):
    """8x FC(lrmul=0.01, use_wscale=True, gain=sqrt 2) -- styleganv1.py:513-518, :532."""
    x = features
    for i in range(layers):
        w = sd[f"{prefix}{i}.weight"]
        w_lrmul, b_lrmul = wscale_fc(w.shape[1], 2 ** 0.5, True, 0.01)
        x = fc(x, w, sd[f"{prefix}{i}.bias"], w_lrmul, b_lrmul)
    return x


def apply_noise(x, weight, noise):
    """``ApplyNoise.forward`` with explicit noise -- styleganv1.py:453-456."""
    return x + weight.view(1, -1, 1, 1) * noise


def style_affine(latent, lin_w, lin_b):
    """The FC inside ``ApplyStyle`` (gain=1, use_wscale=True, lrmul=1) -- styleganv1.py:461,464.
    Returns [B, 2C] (after the FC's LeakyReLU)."""
    w_lrmul, b_lrmul = wscale_fc(lin_w.shape[1], 1.0, True, 1.0)
    return fc(latent, lin_w, lin_b, w_lrmul, b_lrmul)


def apply_style(x, latent, lin_w, lin_b):
    """``ApplyStyle.forward`` -- styleganv1.py:463-468: x*(s0+1)+s1, no normalisation."""
    style = style_affine(latent, lin_w, lin_b).view(-1, 2, x.size(1), 1, 1)
    return x * (style[:, 0] + 1.0) + style[:, 1]


def upsample2x_bilinear(x):
    """``nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)`` -- styleganv1.py:621."""
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


def synthesis_block(x, w2, sd, prefix, noise1, noise2):
    """``SynthesisBlock.forward`` -- styleganv1.py:623-635.  ``w2`` is [B,2,512]."""
    x = upsample2x_bilinear(x)
    x = F.conv2d(x, sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"], padding=1)
    x = apply_noise(x, sd[prefix + "noise1.weight"], noise1)
    x = F.leaky_relu(x, LRELU_SLOPE)
    x = apply_style(x, w2[:, 0], sd[prefix + "style_mod1.linear.weight"], sd[prefix + "style_mod1.linear.bias"])
    x = F.conv2d(x, sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"], padding=1)
    x = apply_noise(x, sd[prefix + "noise2.weight"], noise2)
    x = F.leaky_relu(x, LRELU_SLOPE)
    x = apply_style(x, w2[:, 1], sd[prefix + "style_mod2.linear.weight"], sd[prefix + "style_mod2.linear.bias"])
    return x


def synthesis_num_layers(resolution: int) -> int:
    """``SynthesisNetwork.num_layers`` -- styleganv1.py:572-573."""
    return int(math.log2(resolution)) * 2 - 2


def synthesis_network(w, sd, noises, prefix="synthesis.", resolution: int = 256, return_features=False):
    """``SynthesisNetwork.forward`` -- styleganv1.py:593-610.  ``w`` is [B, L, 512];
    ``noises`` the 2*log2(res)-3 explicit noise tensors in call order."""
    nblocks = int(math.log2(resolution)) - 2
    B = w.size(0)
    x = sd[prefix + "const_input"].expand(B, -1, -1, -1)
    x = x + sd[prefix + "bias"].view(1, -1, 1, 1)
    x = apply_noise(x, sd[prefix + "noise_input1.weight"], noises[0])
    x = apply_style(x, w[:, 0], sd[prefix + "style_mod.linear.weight"], sd[prefix + "style_mod.linear.bias"])
    for i in range(nblocks):
        x = synthesis_block(x, w[:, i * 2 + 1:i * 2 + 3], sd, f"{prefix}layers.{i}.",
                            noises[1 + 2 * i], noises[2 + 2 * i])
    feat = x
    x = F.conv2d(x, sd[prefix + "to_rgb.weight"], sd[prefix + "to_rgb.bias"])
    return (x, feat) if return_features else x


def broadcast_truncate(w512, num_layers: int, truncation_psi=0.7, truncation_cutoff=8):
    """styleganv1.py:536-543: repeat to [B,L,512]; rows [:cutoff] are *scaled* by psi (a plain
    multiply, not a lerp towards a mean latent; applied in train and eval alike)."""
    w = w512.unsqueeze(1).repeat(1, num_layers, 1)
    if truncation_psi and truncation_cutoff:
        coefs = torch.ones_like(w)
        coefs[:, :truncation_cutoff] *= truncation_psi
        w = coefs * w
    return w


def style_generator(features, sd, noises, resolution: int = 256, truncation_psi=0.7,
                    truncation_cutoff=8, mix_features=None, mix_layer=None):
    """``StyleGenerator.forward`` -- styleganv1.py:528-567.

    Eval mode: ``mix_features is None``.  Train-mode style mixing (:547-554) is made explicit:
    the caller passes the ``randn_like(features)`` draw as ``mix_features`` and the
    ``randint(1, L)`` draw as ``mix_layer``; rows ``[mix_layer:]`` are overwritten with the second
    mapping pass, which is *not* truncated (the reference overwrites after truncation).  The
    overwrite is an in-place write under ``no_grad`` (:549-553), so autograd never sees it: in
    backward the gradient of the overwritten rows still flows into the *first* mapping pass.
    That quirk is part of the reference's semantics and is restated as-is.
    """
    L = synthesis_num_layers(resolution)
    w = broadcast_truncate(mapping(features, sd), L, truncation_psi, truncation_cutoff)
    if mix_features is not None:
        with torch.no_grad():
            w2 = mapping(mix_features, sd).unsqueeze(1).repeat(1, L, 1)
            w[:, mix_layer:] = w2[:, mix_layer:]
    return synthesis_network(w, sd, noises, resolution=resolution)


# FLOP accounting used by bench.py's roofline (2*MAC).
def decoder_conv_layers(resolution: int = 256, fmap_base=8192, fmap_max=512):
    """[(cin, cout, out_res, upsampled_input)] for the 3x3 convs, in execution order
    (channel schedule styleganv1.py:575,583-586)."""
    def nf(stage):
        return min(int(fmap_base / (2.0 ** stage)), fmap_max)
    out = []
    for res in range(3, int(math.log2(resolution)) + 1):
        cin, cout, s = nf(res - 2), nf(res - 1), 2 ** res
        out.append((cin, cout, s, True))
        out.append((cout, cout, s, False))
    return out


def decoder_flops_per_frame(resolution: int = 256, input_dim: int = 6144) -> dict:
    conv = sum(2 * ci * co * 9 * s * s for ci, co, s, _ in decoder_conv_layers(resolution))
    layers = decoder_conv_layers(resolution)
    rgb = 2 * layers[-1][1] * 3 * resolution * resolution
    fc_map = 2 * (input_dim * 512 + 7 * 512 * 512)
    style = 2 * 512 * 2 * (512 + sum(co for _, co, _, _ in layers))
    return {"conv3x3": conv, "to_rgb": rgb, "mapping": fc_map, "style": style,
            "total": conv + rgb + fc_map + style}
