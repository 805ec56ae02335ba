"""The opt-in split-precision speed path (include/spk.h SPK_CONV_BF16X3, csrc/conv3x3_bf16x3.hip): 3x3 convs on the bf16
matrix pipe with every operand split into bf16 hi + lo halves and three MFMAs per product into an fp32 accumulator.

The claim to hold it to: fp32-CLASS accuracy -- the operands keep 16 significant bits, products and sums are fp32 -- so a
single layer agrees with an fp64 evaluation to ~1e-5 and the whole decoder stays two orders of magnitude inside the
north-star bound of 1e-3 rel-L2 against the reference's own output (measured here and asserted: < 2e-4)."""
import importlib

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import decoder_ref as R
from oracle import modconv_ref as M
from oracle.weights_recipe import fill_state_dict, recipe_input, recipe_noises, recipe_tensor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    assert torch.cuda.is_available()
    return importlib.import_module("speak-hack_amd")


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 16, 64, 32, 32), (1, 40, 24, 20, 36), (3, 128, 130, 16, 16), (8, 64, 64, 64, 64),
                                             (2, 7, 5, 8, 8), (1, 33, 70, 5, 9), (5, 48, 64, 4, 4)])
def test_plain_conv_vs_fp64(pkg, dev, B, Cin, Cout, H, W):
    """Ragged channels (Cin not a multiple of the 16-channel chunk, Cout off the 64-row tile), odd sizes, planes smaller than a
    tile (several images per tile), bias + LeakyReLU epilogue."""
    ops = pkg.ops
    x = recipe_input(f"bf.x.{B}.{Cin}.{H}.{W}", (B, Cin, H, W))
    w = recipe_tensor(f"bf.w.{Cin}.{Cout}", (Cout, Cin, 3, 3), (9 * Cin) ** -0.5)
    b = recipe_tensor(f"bf.b.{Cout}", (Cout,), 0.3)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.2)
    assert ops.bf16x3_supported(B, Cin, Cout, H, W)
    y = ops.conv3x3_bf16x3(x.to(dev), ops.pack_conv_weight_bf16x3(w.to(dev)), Cout, bias=b.to(dev), lrelu_slope=0.2)
    assert y.shape == ref.shape
    err = rel_l2(y, ref)
    assert err < 3e-5, err                       # 2^-16 operand truncation, fp32 accumulation


@pytest.mark.parametrize("B,Cin,Cout,H", [(2, 32, 64, 16), (1, 20, 40, 9), (8, 128, 64, 32)])
def test_upsample_noise_style_epilogue_vs_oracle(pkg, dev, B, Cin, Cout, H):
    """The decoder's half-block in one launch -- bilinear x2 folded into staging, bias, noise, LeakyReLU, style -- against the
    CPU oracle's op chain (styleganv1.py:624-628) in fp64."""
    ops = pkg.ops
    x = recipe_input(f"bfu.x.{B}.{Cin}.{H}", (B, Cin, H, H))
    w = recipe_tensor(f"bfu.w.{Cin}.{Cout}", (Cout, Cin, 3, 3), (9 * Cin) ** -0.5)
    b, nw = recipe_tensor(f"bfu.b.{Cout}", (Cout,), 0.3), recipe_tensor(f"bfu.nw.{Cout}", (Cout,), 0.2)
    nz = recipe_input(f"bfu.nz.{B}.{H}", (B, 1, 2 * H, 2 * H))
    st = recipe_input(f"bfu.st.{B}.{Cout}", (B, 2 * Cout)) * 0.3
    t = F.conv2d(R.upsample2x_bilinear(x.double()), w.double(), b.double(), padding=1) + nw.double().view(1, -1, 1, 1) * nz.double()
    t = F.leaky_relu(t, 0.2)
    ref = t * (st.double()[:, :Cout, None, None] + 1) + st.double()[:, Cout:, None, None]
    y = ops.conv3x3_bf16x3(x.to(dev), ops.pack_conv_weight_bf16x3(w.to(dev)), Cout, bias=b.to(dev), noise_w=nw.to(dev), noise=nz.to(dev),
                           style=st.to(dev), upsample=True, lrelu_slope=0.2)
    assert rel_l2(y, ref) < 3e-5


def test_decoder_end_to_end_within_the_north_star_bound(pkg, dev, golden):
    """Whole StyleGenerator with ``precision = "bf16x3"``: against the reference's own golden frame (config 1 input) and, at
    the benchmarked size (B = 8, 256^2), against the CPU oracle.  Bound 1e-3 (BASELINE north_star); asserted 2e-4."""
    g = pkg.StyleGenerator(6144).eval()
    sd = fill_state_dict(g.state_dict(), prefix="Gd.")
    g.load_state_dict(sd)
    g.to(dev)
    g.synthesis.precision = "bf16x3"
    try:
        with torch.no_grad():
            y1 = g(recipe_input("e2e.features", (1, 6144)).to(dev), [n.to(dev) for n in recipe_noises("e2e", 1, 256)])
            e1 = rel_l2(y1, golden("decoder_e2e_256.npz")["y"])
            feats, noises = recipe_input("cfg2.features", (8, 6144)), recipe_noises("cfg2", 8, 256)
            y8 = g(feats.to(dev), [n.to(dev) for n in noises])
            ref = R.style_generator(feats, sd, noises)
            e8 = rel_l2(y8, ref)
            plan = next(reversed(g.__dict__["_plans"].values()))
            n_fast = sum(1 for kind, d in plan.ops if kind == pkg._lib.OP_CONV2D and d.flags & pkg._lib.CONV_BF16X3)
    finally:
        g.synthesis.precision = "f32"
    print(f"bf16x3 decoder: rel-L2 {e1:.2e} vs the reference golden (B=1), {e8:.2e} vs the oracle (B=8); {n_fast} of 12 convs on the bf16 pipe")
    # the layers on which the split-precision kernel is the FASTER form: 32^2 and up, except the last conv (fp32 Winograd with toRGB in
    # its epilogue beats bf16x3 + a toRGB pass); the 16^2 layers run quicker -- and exactly -- as sliced fp32 Winograd launches
    assert n_fast >= 7
    assert e1 < 2e-4 and e8 < 2e-4, (e1, e8)
    with torch.no_grad():                         # the default path is untouched: exact fp32
        y_f32 = g(feats.to(dev), [n.to(dev) for n in noises])
    assert rel_l2(y_f32, ref) < 1e-5


def test_stylegan2_variant_bf16x3(pkg, dev):
    """Modulated conv on the split-precision path (modulation in staging, demodulation in the epilogue, upfirdn2d x2 folded in)."""
    sg2 = importlib.import_module("speak-hack_amd.stylegan2")
    torch.manual_seed(5)
    g = sg2.StyleGAN2Generator(6144, resolution=128).eval()
    with torch.no_grad():
        for n, p in g.named_parameters():
            if n.endswith("noise.weight"):
                p.fill_(0.1)
    sd = {k: v.detach().clone() for k, v in g.state_dict().items()}
    B = 8
    feats = recipe_input("sg2.bf.f", (B, 6144))
    noises = [recipe_input(f"sg2.bf.n{i}", s) for i, s in enumerate(M.noise_shapes(B, 128))]
    g.to(dev)
    g.precision = "bf16x3"
    with torch.no_grad():
        ref = M.generator(feats, sd, noises, resolution=128)
        y = g(feats.to(dev), [n.to(dev) for n in noises])
    assert rel_l2(y, ref) < 2e-4


def test_training_switch_decoder_gradients(pkg, dev):
    """``ops.train_conv_precision("bf16x3")``: forward convs and data gradients of the >= 2048-pixel 3x3 layers on the bf16
    pipe, weight gradients exact.  The whole decoder's loss, latent gradient and every parameter gradient stay within the
    north-star bound (1e-3) of the exact path; asserted 3e-4."""
    torch.manual_seed(3)
    g = pkg.StyleGenerator(6144).to(dev).train()
    feats = torch.randn(4, 6144, device=dev, requires_grad=True)
    noises = [torch.randn(s, device=dev) for s in g.synthesis.noise_shapes(4)]
    target = torch.randn(4, 3, 256, 256, device=dev)

    def step():
        g.zero_grad(set_to_none=True)
        feats.grad = None
        y = g(feats, noises, style_mix=False)
        loss = ((y - target) ** 2).mean()
        loss.backward()
        return y.detach().clone(), loss.item(), feats.grad.clone(), {n: p.grad.clone() for n, p in g.named_parameters() if p.grad is not None}

    y0, l0, gf0, gp0 = step()
    with pkg.ops.train_conv_precision("bf16x3"):
        y1, l1, gf1, gp1 = step()
    assert pkg.ops.TRAIN_CONV_PRECISION == "f32"
    assert rel_l2(y1, y0) < 1e-4 and abs(l1 - l0) < 1e-4 * abs(l0)
    assert rel_l2(gf1, gf0) < 3e-4
    errs = sorted(((rel_l2(gp1[n], gp0[n]), n) for n in gp0), reverse=True)
    dense = [e for e in errs if "noise" not in e[1]]
    print(f"bf16x3 training switch: output {rel_l2(y1, y0):.2e}, latent gradient {rel_l2(gf1, gf0):.2e}, worst parameter gradients "
          + ", ".join(f"{e:.1e} {n}" for e, n in errs[:4]) + f"; worst conv / FC / bias gradient {dense[0][0]:.1e} {dense[0][1]}")
    # conv weights, FC weights, biases: the bound.  The per-channel noise weights start at zero, and their gradient
    # sum_{b,p} dt * noise is a sum of 2^16..2^19 random-sign terms that cancels to ~1/sqrt(N) of its terms: the few LeakyReLU
    # masks that flip when a pre-activation moves by 1e-5 show up there first (the same effect as conftest.grad_close describes)
    assert dense[0][0] < 1e-3, dense[0]
    assert errs[0][0] < 2e-2, errs[0]
    assert not torch.equal(y1, y0)                      # the switch did change the arithmetic


def test_training_switch_discriminator_r1(pkg, dev):
    """The discriminator's first-order gradients and its R1 double backward under the training switch against the exact path."""
    torch.manual_seed(5)
    D = importlib.import_module("speak-hack_amd.discriminator").StyleDiscriminator().to(dev).train()
    x = torch.randn(2, 3, 256, 256, device=dev)

    def grads():
        D.zero_grad(set_to_none=True)
        xr = x.clone().requires_grad_(True)
        out = D(xr)
        (gx,) = torch.autograd.grad(out.sum(), xr, create_graph=True)
        loss = torch.nn.functional.softplus(-out).mean() + 5.0 * gx.pow(2).reshape(2, -1).sum(1).mean()
        loss.backward()
        return loss.item(), {n: p.grad.clone() for n, p in D.named_parameters() if p.grad is not None}

    sd = {k: v.clone() for k, v in D.state_dict().items()}          # the power iteration moves u / v: same start for both runs
    l0, g0 = grads()
    D.load_state_dict(sd)
    with pkg.ops.train_conv_precision("bf16x3"):
        l1, g1 = grads()
    assert abs(l1 - l0) < 1e-4 * abs(l0)
    errs = sorted(((rel_l2(g1[n], g0[n]), n) for n in g0), reverse=True)
    med = errs[len(errs) // 2][0]
    print(f"bf16x3 training switch, D + R1: loss {l0:.5f} vs {l1:.5f}; parameter gradients: median {med:.1e}, worst "
          + ", ".join(f"{e:.1e} {n}" for e, n in errs[:4]))
    # first- plus second-order gradients through 14 LeakyReLU layers: a mask that flips when a pre-activation moves by 1e-5
    # shifts every element of the R1 term's gradient a little (conftest.grad_close), most in the first layers
    assert med < 1e-3 and errs[0][0] < 5e-3, errs[:4]
