"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's ``stylegan.py`` generator
(``Generator.forward(w, alpha, steps, zero_noise)``, stylegan.py:159-178, and what it calls) over a
state dict with the reference's keys.  PINNED by tests/golden/progan.npz (tools/make_goldens.py).
Noise is explicit, in the reference's draw order (stylegan.py:81)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

FACTORS = [1, 1, 1, 1, 1 / 2, 1 / 4, 1 / 8, 1 / 16, 1 / 32]


def ws_linear(x, sd, p):
    """WSLinear.forward -- stylegan.py:20-21: linear(x * scale) + bias, scale = sqrt(2 / in)."""
    w = sd[p + "linear.weight"]
    return F.linear(x * (2 / w.shape[1]) ** 0.5, w) + sd[p + "bias"]


def ws_conv(x, sd, p, padding):
    """WSConv2d.forward -- stylegan.py:45-46: conv(x * scale) + bias, scale = sqrt(2 / (in * k^2))."""
    w = sd[p + "conv.weight"]
    scale = (2 / (w.shape[1] * w.shape[2] ** 2)) ** 0.5
    return F.conv2d(x * scale, w, padding=padding) + sd[p + "bias"].view(1, -1, 1, 1)


def pixel_norm(x):
    """stylegan.PixelNorm.forward -- stylegan.py:28-29."""
    return x / torch.sqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)


def mapping_network(z, sd, p="mapping."):
    """MappingNetwork -- stylegan.py:51-71: PixelNorm, then 8 WSLinear with ReLU between."""
    x = pixel_norm(z)
    for i in range(8):
        x = ws_linear(x, sd, f"{p}{1 + 2 * i}.")
        if i < 7:
            x = F.relu(x)
    return x


def adain(x, w, sd, p):
    """AdaIN.forward -- stylegan.py:91-95: nn.InstanceNorm2d (eps 1e-5, no affine) then style scale/bias."""
    x = F.instance_norm(x, eps=1e-5)
    return ws_linear(w, sd, p + "style_scale.")[:, :, None, None] * x + ws_linear(w, sd, p + "style_bias.")[:, :, None, None]


def inject(x, weight, noise):
    return x if noise is None else x + weight * noise


def gen_block(x, w, sd, p, n1, n2):
    """GenBlock.forward -- stylegan.py:108-111."""
    x = adain(F.leaky_relu(inject(ws_conv(x, sd, p + "conv1.", 1), sd[p + "inject_noise1.weight"], n1), 0.2), w, sd, p + "adain1.")
    return adain(F.leaky_relu(inject(ws_conv(x, sd, p + "conv2.", 1), sd[p + "inject_noise2.weight"], n2), 0.2), w, sd, p + "adain2.")


def generator(w, alpha, steps, sd, noises=None):
    """Generator.forward -- stylegan.py:159-178.  ``noises`` None = zero_noise.

    Quirk restated as-is: ``self.leaky`` is in-place (stylegan.py:136); with zero_noise ``InjectNoise`` returns
    its input (stylegan.py:79-80), so line 162 overwrites ``x`` and ``initial_rgb(x)`` at steps == 0 sees the
    LeakyReLU'd tensor; with noise the add makes a copy and ``x`` stays the raw conv output."""
    nz = iter(noises) if noises is not None else None
    nxt = (lambda: next(nz)) if nz is not None else (lambda: None)
    x = adain(inject(sd["starting_constant"], sd["initial_noise1.weight"], nxt()), w, sd, "initial_adain1.")
    x = F.conv2d(x, sd["initial_conv.weight"], sd["initial_conv.bias"], padding=1)
    pre = F.leaky_relu(inject(x, sd["initial_noise2.weight"], nxt()), 0.2)
    if noises is None:
        x = pre
    out = adain(pre, w, sd, "initial_adain2.")
    if steps == 0:
        return ws_conv(x, sd, "initial_rgb.", 0)
    for step in range(steps):
        upscaled = F.interpolate(out, scale_factor=2, mode="bilinear")
        out = gen_block(upscaled, w, sd, f"prog_blocks.{step}.", nxt(), nxt())
    final_upscaled = ws_conv(upscaled, sd, f"rgb_layers.{steps - 1}.", 0)
    final_out = ws_conv(out, sd, f"rgb_layers.{steps}.", 0)
    return torch.tanh(alpha * final_out + (1 - alpha) * final_upscaled)       # fade_in, stylegan.py:155-157


def generator_param_shapes(w_dim=512, in_channels=512, img_channels=3):
    """key -> shape of ``Generator(w_dim, in_channels)``'s state dict (145 entries; ``rgb_layers.0`` aliases
    ``initial_rgb`` exactly as the reference's shared module does)."""
    sd = {"starting_constant": (1, in_channels, 4, 4)}

    def adain_keys(p, c):
        for s in ("style_scale", "style_bias"):
            sd[f"{p}.{s}.bias"] = (c,)
            sd[f"{p}.{s}.linear.weight"] = (c, w_dim)

    adain_keys("initial_adain1", in_channels)
    adain_keys("initial_adain2", in_channels)
    sd["initial_noise1.weight"] = (1, in_channels, 1, 1)
    sd["initial_noise2.weight"] = (1, in_channels, 1, 1)
    sd["initial_conv.weight"] = (in_channels, in_channels, 3, 3)
    sd["initial_conv.bias"] = (in_channels,)
    sd["initial_rgb.bias"] = (img_channels,)
    sd["initial_rgb.conv.weight"] = (img_channels, in_channels, 1, 1)
    for i in range(len(FACTORS) - 1):
        cin, cout = int(in_channels * FACTORS[i]), int(in_channels * FACTORS[i + 1])
        p = f"prog_blocks.{i}"
        sd[f"{p}.conv1.bias"] = (cout,)
        sd[f"{p}.conv1.conv.weight"] = (cout, cin, 3, 3)
        sd[f"{p}.conv2.bias"] = (cout,)
        sd[f"{p}.conv2.conv.weight"] = (cout, cout, 3, 3)
        sd[f"{p}.inject_noise1.weight"] = (1, cout, 1, 1)
        sd[f"{p}.inject_noise2.weight"] = (1, cout, 1, 1)
        adain_keys(f"{p}.adain1", cout)
        adain_keys(f"{p}.adain2", cout)
    sd["rgb_layers.0.bias"] = (img_channels,)
    sd["rgb_layers.0.conv.weight"] = (img_channels, in_channels, 1, 1)
    for i in range(len(FACTORS) - 1):
        cout = int(in_channels * FACTORS[i + 1])
        sd[f"rgb_layers.{i + 1}.bias"] = (img_channels,)
        sd[f"rgb_layers.{i + 1}.conv.weight"] = (img_channels, cout, 1, 1)
    return sd


def generator_recipe_state_dict():
    """The weights tools/make_goldens.py gave the reference Generator: N(0,1) WS weights (the module scales its
    input), He-scaled plain ``initial_conv``; ``rgb_layers.0`` and ``initial_rgb`` share one tensor -- the
    reference's load_state_dict copies ``initial_rgb.*`` then ``rgb_layers.0.*`` into the same storage, so the
    LATER key (rgb_layers.0) wins."""
    from .weights_recipe import recipe_tensor
    sd = {}
    for k, shp in generator_param_shapes().items():
        if k == "initial_conv.weight":
            sd[k] = recipe_tensor("progan." + k, shp)
        elif k.endswith("weight") and len(shp) in (2, 4):       # WS weights AND the [1,C,1,1] noise weights
            sd[k] = recipe_tensor("progan." + k, shp, 1.0)
        else:
            sd[k] = recipe_tensor("progan." + k, shp)
    sd["initial_rgb.bias"], sd["initial_rgb.conv.weight"] = sd["rgb_layers.0.bias"], sd["rgb_layers.0.conv.weight"]
    return sd


def noise_shapes(B, steps):
    shapes = [(1, 1, 4, 4), (B, 1, 4, 4)]          # initial_noise1 sees the un-expanded constant
    for s in range(steps):
        r = 8 * 2 ** s
        shapes += [(B, 1, r, r), (B, 1, r, r)]
    return shapes
