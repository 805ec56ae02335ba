"""TEST INFRASTRUCTURE ONLY -- deterministic, constructor-order-independent weight recipe.

Goldens never store weights: both the golden generator (which fills the *reference's* modules)
and the tests (which fill the build's modules / the oracle's state dict) regenerate every
tensor from its ``state_dict`` key: ``np.random.RandomState(crc32(key))`` scaled per key kind.
``RandomState`` streams are stable across numpy versions.

The scales keep activations O(1) through the 13 stacked layers and -- unlike the reference's
default init, where ``ApplyNoise.weight`` is all zeros (styleganv1.py:451) and every bias is
zero -- exercise the noise and bias paths.
"""
from __future__ import annotations

import re
import zlib

import numpy as np
import torch


def _rs(key: str) -> np.random.RandomState:
    return np.random.RandomState(zlib.crc32(key.encode("utf-8")) & 0xFFFFFFFF)


def _scale_for(key: str, shape) -> float:
    """Per-key standard deviation (keys are matched on their *suffix*, prefixes are free)."""
    if re.search(r"mapping\.\d+\.weight$", key):
        return 100.0                      # FC(lrmul=0.01, use_wscale): init_std = 1/lrmul
    if re.search(r"mapping\.\d+\.bias$", key):
        return 10.0                       # effective bias = 0.01 * this
    if key.endswith("linear.weight"):
        return 1.0                        # style FC, use_wscale, lrmul=1
    if key.endswith("linear.bias"):
        return 0.1
    if key.endswith("const_input") or key.endswith("starting_constant"):
        return 1.0
    if re.search(r"noise\w*\.weight$", key):
        return 0.1
    if key.endswith("bias"):
        return 0.1
    if key.endswith("weight") and len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        return float(np.sqrt(2.0 / fan_in))
    if key.endswith("weight") and len(shape) == 2:
        return float(np.sqrt(2.0 / shape[1]))
    if key.endswith("weight") and len(shape) == 1:
        return 0.1
    return 1.0


def recipe_tensor(key: str, shape, scale: float | None = None) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    a = _rs(key).standard_normal(shape).astype(np.float32)
    s = _scale_for(key, shape) if scale is None else scale
    return torch.from_numpy(a * np.float32(s))


def fill_state_dict(sd: dict, prefix: str = "", wscale_convs: bool = False) -> dict:
    """Return a new dict with every floating tensor of ``sd`` replaced by its recipe tensor.

    ``prefix`` is prepended to the key before hashing so that two modules with the same local
    key layout (e.g. three encoders) get different weights.  ``wscale_convs``: stylegan.py's
    WSConv2d / WSLinear keep N(0,1) weights and scale the *input* instead (stylegan.py:12,37).
    """
    out = {}
    for k, v in sd.items():
        if not torch.is_floating_point(v):
            out[k] = v.clone()
            continue
        if wscale_convs and k.endswith("weight") and v.dim() in (2, 4):
            out[k] = recipe_tensor(prefix + k, v.shape, 1.0)
        else:
            out[k] = recipe_tensor(prefix + k, v.shape)
    return out


def resnet_trunk_state_dict(prefix: str) -> dict:
    """Recipe state dict of one ResNet-50 trunk (torchvision Sequential keys, oracle.resnet_ref):
    conv weights He-scaled, BN gamma = 1 + 0.2 N, beta = 0.1 N, running_mean = 0.1 N,
    running_var = 0.5 + |N| (positive), num_batches_tracked = 0."""
    from .resnet_ref import trunk_param_shapes
    sd = {}
    for k, shp in trunk_param_shapes().items():
        key = prefix + k
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_var"):
            sd[k] = recipe_tensor(key, shp, 1.0).abs() + 0.5
        elif k.endswith("running_mean"):
            sd[k] = recipe_tensor(key, shp, 0.1)
        elif len(shp) == 1 and k.endswith("weight"):
            sd[k] = 1.0 + recipe_tensor(key, shp, 0.2)
        elif len(shp) == 1:
            sd[k] = recipe_tensor(key, shp, 0.1)
        else:
            sd[k] = recipe_tensor(key, shp)          # conv: sqrt(2 / fan_in)
    return sd


def recipe_input(name: str, shape, dist: str = "normal") -> torch.Tensor:
    rs = _rs("input:" + name)
    if dist == "normal":
        a = rs.standard_normal(tuple(shape))
    elif dist == "uniform":               # U(-1, 1): image-like inputs (train.py:374-379)
        a = rs.uniform(-1.0, 1.0, tuple(shape))
    else:
        raise ValueError(dist)
    return torch.from_numpy(a.astype(np.float32))


def decoder_noise_shapes(batch: int, resolution: int = 256):
    """Shapes of the 2*log2(res)-3 noise draws of one SynthesisNetwork.forward, in call order
    (styleganv1.py:598 then :626,:631 per block)."""
    log2 = int(np.log2(resolution))
    shapes = [(batch, 1, 4, 4)]
    for res in range(3, log2 + 1):
        s = 2 ** res
        shapes += [(batch, 1, s, s), (batch, 1, s, s)]
    return shapes


def recipe_noises(tag: str, batch: int, resolution: int = 256):
    return [recipe_input(f"{tag}:noise{i}", s) for i, s in enumerate(decoder_noise_shapes(batch, resolution))]
