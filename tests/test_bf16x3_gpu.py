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
    assert n_fast >= 10                           # every layer from 16^2 up at B = 8
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
