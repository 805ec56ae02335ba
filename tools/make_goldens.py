#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own decoder code.

Runs only in the build container (needs /root/reference; never on the GPU box).  It imports
``/root/reference/styleganv1.py`` and ``/root/reference/stylegan.py`` as they lie there (a stub
module stands in for the *unused* ``import torchvision.utils`` at styleganv1.py:24), fills every
parameter from ``oracle.weights_recipe`` (weights are never stored), replaces the
``noise is None`` draw of ``ApplyNoise.forward`` / ``InjectNoise.forward`` by recipe noise so the
outputs are reproducible, and stores inputs + expected outputs (+ selected gradients).

    python tools/make_goldens.py            # writes tests/golden/*.npz

Fixtures are data only (inputs / outputs); no reference source text is stored.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle.weights_recipe import (fill_state_dict, recipe_input, recipe_noises, recipe_tensor)  # noqa: E402


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; goldens are generated in the build container only")
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tv.utils = tvu
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.utils", tvu)
    sys.path.insert(0, REF)
    import stylegan as ref_progan          # noqa
    import styleganv1 as ref_sg            # noqa
    return ref_sg, ref_progan


class NoiseFeeder:
    """Feeds recorded noise tensors to the reference's noise modules in call order."""

    def __init__(self):
        self.queue = []

    def load(self, noises):
        self.queue = list(noises)

    def pop(self, shape):
        n = self.queue.pop(0)
        assert tuple(n.shape) == tuple(shape), (n.shape, shape)
        return n


def patch_noise(ref_sg, ref_progan, feeder):
    orig = ref_sg.ApplyNoise.forward

    def apply_noise_forward(self, x, noise):
        if noise is None:
            noise = feeder.pop((x.size(0), 1, x.size(2), x.size(3)))
        return orig(self, x, noise)

    ref_sg.ApplyNoise.forward = apply_noise_forward

    def inject_forward(self, x, zero_noise=False):
        if zero_noise:
            return x
        noise = feeder.pop((x.shape[0], 1, x.shape[2], x.shape[3]))
        return x + self.weight * noise

    ref_progan.InjectNoise.forward = inject_forward


def npf(t):
    return t.detach().cpu().numpy().astype(np.float32)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"  {name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(arrays)} arrays")


def strided(t, s):
    return t[..., ::s, ::s]


# ----------------------------------------------------------------------------------------------
def gen_ops(ref_sg):
    """G1: per-op fixtures incl. gradients."""
    out = {}
    # FC, three flavours: mapping-style, style-affine-style, no-wscale/no-bias
    for tag, kw, shp in [("fc_map", dict(lrmul=0.01, use_wscale=True), (3, 40, 24)),
                         ("fc_style", dict(gain=1.0, use_wscale=True), (3, 24, 32)),
                         ("fc_plain", dict(gain=2 ** 0.5, use_wscale=False, bias=False), (2, 17, 9))]:
        B, I, O = shp
        m = ref_sg.FC(I, O, **kw)
        m.load_state_dict(fill_state_dict(m.state_dict(), prefix=tag + "."))
        x = recipe_input(tag + ".x", (B, I)).requires_grad_(True)
        y = m(x)
        gy = recipe_input(tag + ".gy", y.shape)
        y.backward(gy)
        out.update({f"{tag}.x": npf(x), f"{tag}.y": npf(y), f"{tag}.gy": npf(gy), f"{tag}.gx": npf(x.grad),
                    f"{tag}.gw": npf(m.weight.grad)})
        if m.bias is not None:
            out[f"{tag}.gb"] = npf(m.bias.grad)
    # ApplyNoise (explicit noise) and ApplyStyle
    C, B, H = 6, 2, 5
    m = ref_sg.ApplyNoise(C)
    m.load_state_dict(fill_state_dict(m.state_dict(), prefix="an."))
    x = recipe_input("an.x", (B, C, H, H)).requires_grad_(True)
    nz = recipe_input("an.noise", (B, 1, H, H))
    y = m(x, nz)
    gy = recipe_input("an.gy", y.shape)
    y.backward(gy)
    out.update({"an.x": npf(x), "an.noise": npf(nz), "an.y": npf(y), "an.gy": npf(gy), "an.gx": npf(x.grad),
                "an.gw": npf(m.weight.grad)})
    m = ref_sg.ApplyStyle(16, C, use_wscale=True)
    m.load_state_dict(fill_state_dict(m.state_dict(), prefix="as."))
    x = recipe_input("as.x", (B, C, H, H)).requires_grad_(True)
    lat = recipe_input("as.lat", (B, 16)).requires_grad_(True)
    y = m(x, lat)
    gy = recipe_input("as.gy", y.shape)
    y.backward(gy)
    out.update({"as.x": npf(x), "as.lat": npf(lat), "as.y": npf(y), "as.gy": npf(gy), "as.gx": npf(x.grad),
                "as.glat": npf(lat.grad), "as.gw": npf(m.linear.weight.grad), "as.gb": npf(m.linear.bias.grad)})
    # nn.Upsample exactly as SynthesisBlock builds it (styleganv1.py:621)
    up = torch.nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)
    for tag, shp in [("up_a", (2, 3, 4, 4)), ("up_b", (1, 2, 7, 5)), ("up_c", (1, 1, 1, 1))]:
        x = recipe_input(tag + ".x", shp).requires_grad_(True)
        y = up(x)
        gy = recipe_input(tag + ".gy", y.shape)
        y.backward(gy)
        out.update({f"{tag}.x": npf(x), f"{tag}.y": npf(y), f"{tag}.gy": npf(gy), f"{tag}.gx": npf(x.grad)})
    save("decoder_ops.npz", **out)


def gen_blocks(ref_sg, feeder):
    """G2: SynthesisBlock at real channel counts, reduced spatial size; fwd + grads."""
    out = {}
    for tag, cin, cout, B, hin in [("blk512", 512, 512, 2, 4), ("blk128_64", 128, 64, 1, 16),
                                   ("blk16_8", 16, 8, 3, 6)]:
        m = ref_sg.SynthesisBlock(cin, cout, 3)
        m.load_state_dict(fill_state_dict(m.state_dict(), prefix=tag + "."))
        x = recipe_input(tag + ".x", (B, cin, hin, hin)).requires_grad_(True)
        w = recipe_input(tag + ".w", (B, 2, 512)).requires_grad_(True)
        n1 = recipe_input(tag + ".n1", (B, 1, 2 * hin, 2 * hin))
        n2 = recipe_input(tag + ".n2", (B, 1, 2 * hin, 2 * hin))
        feeder.load([n1, n2])
        y = m(x, w)
        gy = recipe_input(tag + ".gy", y.shape)
        y.backward(gy)
        out.update({f"{tag}.y": npf(y), f"{tag}.gx": npf(x.grad), f"{tag}.gw": npf(w.grad)})
        for pn, p in m.named_parameters():
            g = p.grad
            if g.dim() == 4:                                  # conv weights: slice + norm only
                out[f"{tag}.g.{pn}.slice"] = npf(g[:8, :8])
                out[f"{tag}.g.{pn}.norm"] = np.float64(g.double().norm().item())
            elif g.numel() > 4096:
                out[f"{tag}.g.{pn}.slice"] = npf(g[:8, :64])
                out[f"{tag}.g.{pn}.norm"] = np.float64(g.double().norm().item())
            else:
                out[f"{tag}.g.{pn}"] = npf(g)
    save("decoder_blocks.npz", **out)


def gen_e2e(ref_sg, feeder):
    """G3: whole StyleGenerator, eval, B=1, full 256^2 output; B=2 strided."""
    g = ref_sg.StyleGenerator(6144).eval()
    g.load_state_dict(fill_state_dict(g.state_dict(), prefix="Gd."))
    feats = recipe_input("e2e.features", (1, 6144))
    feeder.load(recipe_noises("e2e", 1, 256))
    with torch.no_grad():
        y = g(feats)
    save("decoder_e2e_256.npz", y=npf(y), norm=np.float64(y.double().norm().item()))

    feats = recipe_input("e2e_b2.features", (2, 6144))
    feeder.load(recipe_noises("e2e_b2", 2, 256))
    with torch.no_grad():
        w = g.mapping(feats)
        y = g(feats)
    save("decoder_e2e_256_b2.npz", w=npf(w), y_s4=npf(strided(y, 4)), y_crop=npf(y[..., 96:160, 96:160]),
         norm=np.float64(y.double().norm().item()))

    # G4: train-mode style mixing with a fixed host seed; draws replayed to record them
    g.train()
    seed = 1234
    feats = recipe_input("mix.features", (1, 6144))
    torch.manual_seed(seed)
    r = torch.rand(1)
    mix_features = torch.randn_like(feats)
    mix_layer = int(torch.randint(1, g.synthesis.num_layers, (1,)).item())
    assert float(r) < g.style_mixing_prob, "pick a seed that takes the mixing branch"
    feats_g = feats.clone().requires_grad_(True)
    torch.manual_seed(seed)
    feeder.load(recipe_noises("mix", 1, 256))
    y = g(feats_g)
    gy = recipe_input("mix.gy", y.shape)
    g.zero_grad()
    y.backward(gy)
    save("decoder_train_mix.npz", mix_features=npf(mix_features), mix_layer=np.int64(mix_layer),
         seed=np.int64(seed), y_s4=npf(strided(y, 4)), norm=np.float64(y.double().norm().item()),
         gfeat=npf(feats_g.grad),
         g_map7_bias=npf(g.mapping[7].bias.grad), g_const=npf(g.synthesis.const_input.grad),
         g_rgb_w=npf(g.synthesis.to_rgb.weight.grad), g_rgb_b=npf(g.synthesis.to_rgb.bias.grad),
         g_l5_noise2=npf(g.synthesis.layers[5].noise2.weight.grad),
         g_l5_conv2_b=npf(g.synthesis.layers[5].conv2.bias.grad),
         g_l5_conv2_w=npf(g.synthesis.layers[5].conv2.weight.grad),
         g_l0_conv1_w_slice=npf(g.synthesis.layers[0].conv1.weight.grad[:8, :8]),
         g_l0_conv1_w_norm=np.float64(g.synthesis.layers[0].conv1.weight.grad.double().norm().item()),
         g_l3_style1_b=npf(g.synthesis.layers[3].style_mod1.linear.bias.grad))
    g.eval()

    # 512^2 synthesis-only (config 5): 16 w rows, crops + strided sample
    s = ref_sg.SynthesisNetwork(resolution=512).eval()
    s.load_state_dict(fill_state_dict(s.state_dict(), prefix="Gd512.synthesis."))
    w = recipe_input("e2e512.w", (1, s.num_layers, 512))
    feeder.load(recipe_noises("e2e512", 1, 512))
    with torch.no_grad():
        y = s(w)
    save("decoder_e2e_512.npz", y_s8=npf(strided(y, 8)), y_crop=npf(y[..., 224:288, 224:288]),
         norm=np.float64(y.double().norm().item()))


def gen_legacy(ref_sg, ref_progan):
    out = {}
    x = recipe_input("legacy.x", (2, 6, 9, 7)).requires_grad_(True)
    for tag, fn in [("pixelnorm", ref_sg.PixelNorm()), ("instnorm", ref_sg.InstanceNorm()),
                    ("blur", ref_sg.Blur2d()), ("blur_s2", ref_sg.Blur2d(stride=2)),
                    ("blur_flip", ref_sg.Blur2d(f=[1, 2, 3], flip=True)),
                    ("upscale", ref_sg.Upscale2d()), ("upscale_g", ref_sg.Upscale2d(factor=2, gain=0.5)),
                    ("pixelnorm_sqrt", ref_progan.PixelNorm())]:
        x.grad = None
        y = fn(x)
        gy = recipe_input(f"legacy.{tag}.gy", y.shape)
        y.backward(gy)
        out.update({f"{tag}.y": npf(y), f"{tag}.gy": npf(gy), f"{tag}.gx": npf(x.grad)})
    out["x"] = npf(x)
    save("legacy_ops.npz", **out)


def gen_fused_upscale(ref_sg):
    """F3: the fused ``ConvTranspose2d(4, stride 2, pad 1)`` upscale the reference's ``GBlock`` builds for res >= 7
    (styleganv1.py:231), taken from a constructed ``GBlock`` (real channel counts nf(4)=512 -> nf(5)=256, small ragged image)."""
    blk = ref_sg.GBlock(7, True, True, False, True, noise_input=None)
    up = blk.up_sample
    assert isinstance(up, torch.nn.ConvTranspose2d) and tuple(up.weight.shape) == (512, 256, 4, 4)
    with torch.no_grad():
        up.weight.copy_(recipe_tensor("legacy.fused_upscale.weight", up.weight.shape, 1.0) * (512 * 4) ** -0.5)
        up.bias.copy_(recipe_tensor("legacy.fused_upscale.bias", up.bias.shape, 0.5))
        x = recipe_input("legacy.fused_upscale.x", (2, 512, 8, 6))
        y = up(x)
    # round 3: the backward of the same module (autograd of the reference's own ConvTranspose2d): gradient w.r.t. the
    # input in full, w.r.t. the bias in full, w.r.t. the [512,256,4,4] weight as every 8th input x every 4th output channel
    # (all 16 taps) plus its norm -- 131 KB instead of 8 MB
    gy = recipe_input("legacy.fused_upscale.gy", tuple(y.shape))
    xg = x.clone().requires_grad_(True)
    up.zero_grad()
    up(xg).backward(gy)
    gw = up.weight.grad
    # (gy is not stored: the tests regenerate it from the same recipe key)
    save("legacy_fused_upscale.npz", x=npf(x), y=npf(y), gx=npf(xg.grad), gb=npf(up.bias.grad),
         gw_sample=npf(gw[::8, ::4]), gw_norm=np.float64(gw.double().norm().item()))


def progan_noise_shapes(B, steps):
    shapes = [(1, 1, 4, 4), (B, 1, 4, 4)]        # initial_noise1 sees the un-expanded constant
    for s in range(steps):
        r = 8 * 2 ** s
        shapes += [(B, 1, r, r), (B, 1, r, r)]
    return shapes


def gen_progan(ref_progan, feeder):
    """G5: stylegan.Generator(w, alpha, steps, zero_noise) -- stylegan.py:159-178."""
    g = ref_progan.Generator(512, 512).eval()
    sd = fill_state_dict(g.state_dict(), prefix="progan.", wscale_convs=True)
    # initial_conv is a plain nn.Conv2d (stylegan.py:135): give it a fan-in scaled weight
    sd["initial_conv.weight"] = recipe_tensor("progan.initial_conv.weight", sd["initial_conv.weight"].shape)
    g.load_state_dict(sd)
    out = {}
    for steps, alpha, zero_noise, B in [(0, 1.0, True, 2), (3, 0.3, True, 2), (3, 1.0, False, 2), (6, 0.3, False, 1)]:
        tag = f"s{steps}_a{alpha}_z{int(zero_noise)}"
        w = recipe_input(f"progan.{tag}.w", (B, 512))
        if not zero_noise:
            feeder.load([recipe_input(f"progan.{tag}.n{i}", s) for i, s in enumerate(progan_noise_shapes(B, steps))])
        with torch.no_grad():
            y = g(w, alpha, steps, zero_noise)
        out[f"{tag}.y"] = npf(y if y.shape[-1] <= 64 else strided(y, 4))
        out[f"{tag}.norm"] = np.float64(y.double().norm().item())
    save("progan.npz", **out)


def main():
    torch.set_num_threads(8)
    torch.manual_seed(0)
    os.makedirs(OUT, exist_ok=True)
    ref_sg, ref_progan = import_reference()
    feeder = NoiseFeeder()
    patch_noise(ref_sg, ref_progan, feeder)
    print("generating goldens from", REF)
    if "--only-fused-upscale" in sys.argv:     # added in round 2: leaves the other fixtures untouched
        gen_fused_upscale(ref_sg)
        return
    gen_fused_upscale(ref_sg)
    gen_ops(ref_sg)
    gen_blocks(ref_sg, feeder)
    gen_e2e(ref_sg, feeder)
    gen_legacy(ref_sg, ref_progan)
    gen_progan(ref_progan, feeder)


if __name__ == "__main__":
    main()
