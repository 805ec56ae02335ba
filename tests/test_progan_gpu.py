"""GPU parity of the stylegan.py generator mirror (A12) and of the stand-alone legacy ops (PixelNorm x2,
InstanceNorm, Blur2d, Upscale2d) against the reference's golden vectors."""
import importlib

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import progan_ref as P
from oracle.weights_recipe import recipe_input
from test_progan_oracle import CASES, case_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_generator_vs_reference_goldens(dev, golden):
    import stylegan                                     # the top-level drop-in
    g = golden("progan.npz")
    gen = stylegan.Generator(512, 512).eval()
    sd = P.generator_recipe_state_dict()
    assert set(gen.state_dict().keys()) == set(sd.keys()) and len(sd) == 145
    assert {k: tuple(v.shape) for k, v in gen.state_dict().items()} == P.generator_param_shapes()
    gen.load_state_dict(sd)
    gen.to(dev)
    for steps, alpha, zero_noise, B in CASES:
        tag, w, noises = case_inputs(steps, alpha, zero_noise, B)
        with torch.no_grad():
            y = gen(w.to(dev), alpha, steps, zero_noise, None if noises is None else [n.to(dev) for n in noises])
        assert y.shape == (B, 3, 4 * 2 ** steps, 4 * 2 ** steps)
        got = y if y.shape[-1] <= 64 else y[..., ::4, ::4]
        assert rel_l2(got, g[f"{tag}.y"]) < 2e-4, tag          # tanh-saturated outputs; 2*steps+1 stacked layers
    with torch.no_grad():                                   # device-drawn noise path runs
        assert torch.isfinite(gen(w.to(dev), 0.5, 2)).all()
    with pytest.raises(NotImplementedError):
        gen(w.to(dev), 0.5, 1, True)                        # grad mode: backward not built for this orphan module


def test_mapping_network_vs_oracle(dev):
    import stylegan
    m = stylegan.MappingNetwork(512, 512).eval()
    z = recipe_input("progan.map.z", (3, 512))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = P.mapping_network(z, sd)
        out = m.to(dev)(z.to(dev))
    assert rel_l2(out, ref) < 2e-5


def test_legacy_ops_vs_reference_goldens(dev, golden):
    g = golden("legacy_ops.npz")
    lg = importlib.import_module("speak-hack_amd.legacy")
    import stylegan
    x = T(g["x"]).to(dev)
    mods = {"pixelnorm": lg.PixelNorm(), "instnorm": lg.InstanceNorm(), "blur": lg.Blur2d(), "blur_s2": lg.Blur2d(stride=2),
            "blur_flip": lg.Blur2d(f=[1, 2, 3], flip=True), "upscale": lg.Upscale2d(), "upscale_g": lg.Upscale2d(factor=2, gain=0.5),
            "pixelnorm_sqrt": stylegan.PixelNorm()}
    for tag, mod in mods.items():
        with torch.no_grad():
            y = mod(x)
        assert rel_l2(y, g[f"{tag}.y"]) < 2e-6, tag
    # StyleGAN2's [1,3,3,1] FIR (4x4 taps) through the same kernel, vs a depthwise conv
    f = torch.tensor([1., 3., 3., 1.])
    k2 = (f[:, None] * f[None, :]) / 64.0
    xc = T(g["x"])
    ref = torch.nn.functional.conv2d(xc, k2[None, None].expand(xc.size(1), -1, -1, -1), padding=1, groups=xc.size(1))
    ops = importlib.import_module("speak-hack_amd.ops")
    assert rel_l2(ops.blur2d(x, k2), ref) < 2e-6
