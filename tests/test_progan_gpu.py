"""GPU parity of the stylegan.py generator mirror (A12) and of the stand-alone legacy ops (PixelNorm x2,
InstanceNorm, Blur2d, Upscale2d) against the reference's golden vectors."""
import importlib

import numpy as np
import pytest
import torch

from conftest import grad_close, rel_l2
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


@pytest.mark.parametrize("steps,alpha", [(0, 1.0), (2, 0.3), (3, 1.0)])
def test_generator_backward_vs_oracle_autograd(dev, steps, alpha):
    """A12 backward: mapping network + generator, gradients of a quadratic loss w.r.t. every parameter and z, against
    autograd of the oracle (which is pinned to the reference's forward by the goldens) in fp64."""
    prog = importlib.import_module("speak-hack_amd.progan")
    B = 2
    torch.manual_seed(7)
    g, mp = prog.Generator(512, 512).train(), prog.MappingNetwork(512, 512).train()
    with torch.no_grad():
        for n, p in g.named_parameters():
            if n.endswith("noise1.weight") or n.endswith("noise2.weight"):
                p.normal_(0, 0.3)
            elif n.endswith("bias"):
                p.normal_(0, 0.1)
    z = recipe_input(f"pgb.z.{B}", (B, 512))
    noises = [recipe_input(f"pgb.n{i}.{steps}", s) for i, s in enumerate(P.noise_shapes(B, steps))]
    res = 4 * 2 ** steps
    target = recipe_input(f"pgb.t.{steps}", (B, 3, res, res))
    out = {}
    for name, dt_, device in (("ref32", torch.float32, "cpu"), ("ref64", torch.float64, "cpu"), ("hip", torch.float32, dev)):
        zi = z.detach().clone().to(device, dt_).requires_grad_(True)
        if name == "hip":
            gg, mm = g.to(dev), mp.to(dev)
            y = gg(mm(zi), alpha, steps, noises=[n.to(dev) for n in noises])
            params = {"g." + k: v for k, v in gg.named_parameters()}
            params.update({"m." + k: v for k, v in mm.named_parameters()})
        else:
            gsd = {k: v.detach().clone().to(dt_).requires_grad_(True) for k, v in g.state_dict().items()}
            msd = {k: v.detach().clone().to(dt_).requires_grad_(True) for k, v in mp.state_dict().items()}
            y = P.generator(P.mapping_network(zi, msd, p="mapping."), alpha, steps, gsd, [n.to(dt_) for n in noises])
            params = {"g." + k: v for k, v in gsd.items()}
            params.update({"m." + k: v for k, v in msd.items()})
        ((y - target.to(device, dt_)) ** 2).mean().backward()
        gr = {k: p.grad for k, p in params.items() if p.grad is not None}
        gr["z"] = zi.grad
        out[name] = (y, gr)
    assert rel_l2(out["hip"][0], out["ref64"][0]) < 2e-4
    ref_keys = {k for k, v in out["ref64"][1].items() if float(v.abs().max()) > 0}
    hip = out["hip"][1]
    # rgb_layers.0 aliases initial_rgb (one Parameter, two names): the module reports it once
    ref_keys = {k for k in ref_keys if not k.startswith("g.rgb_layers.0.")} if steps else ref_keys
    missing = {k for k in ref_keys if k not in hip and k.replace("g.rgb_layers.0.", "g.initial_rgb.") not in hip}
    assert not missing, sorted(missing)[:6]
    for k in sorted(ref_keys):
        kk = k if k in hip else k.replace("g.rgb_layers.0.", "g.initial_rgb.")
        ok, info = grad_close(hip[kk], out["ref32"][1][k], out["ref64"][1][k])
        assert ok, (k, info)
