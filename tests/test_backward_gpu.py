"""GPU parity of the backward path (decoder part): every backward kernel against
PyTorch autograd on the CPU oracle, the SynthesisBlock gradients against the reference's own
(golden G2) and the whole StyleGenerator train-mode step against golden G4 (incl. the style-mixing
gradient quirk).  Tolerances: 2e-5 rel-L2 per kernel, 1e-4 per block, 5e-4 end to end (gradients
flow back through 13 layers; exact fp32 arithmetic, summation order differs)."""
import importlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import decoder_ref as R
from oracle.weights_recipe import fill_state_dict, recipe_input, recipe_noises, recipe_tensor

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def pkg():
    assert torch.cuda.is_available()
    p = importlib.import_module("speak-hack_amd")
    p._lib.lib()
    return p


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("k,stride,B,Cin,Cout,H,W", [
    (3, 1, 2, 64, 64, 32, 32), (3, 1, 3, 20, 40, 9, 13), (3, 1, 1, 128, 72, 64, 64), (3, 1, 8, 512, 512, 8, 8),
    (1, 1, 2, 96, 160, 16, 16), (1, 2, 2, 64, 128, 16, 16), (3, 2, 2, 48, 80, 20, 20), (7, 2, 2, 3, 64, 40, 40),
    (3, 1, 1, 5, 3, 2, 2), (1, 1, 2, 2048, 512, 1, 1),
    # the pipelined 3x3 form (16x4 tiles): partial tiles in both directions, channel blocks with 6 / 2 live rows
    (3, 1, 3, 24, 40, 6, 20), (3, 1, 2, 70, 130, 12, 36),
])
def test_wgrad_vs_autograd(pkg, dev, k, stride, B, Cin, Cout, H, W):
    tag = f"wg.{k}.{stride}.{B}.{Cin}.{Cout}.{H}.{W}"
    x = recipe_input(tag + ".x", (B, Cin, H, W))
    w = recipe_tensor(tag + ".weight", (Cout, Cin, k, k)).requires_grad_(True)
    y = F.conv2d(x, w, stride=stride, padding=(k - 1) // 2)
    g = recipe_input(tag + ".g", y.shape)
    y.backward(g)
    for splits in (0, 1, 3):
        dw = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), Cout, Cin, k, stride, splits=splits)
        assert rel_l2(dw, w.grad) < TOL, splits
    # accumulate + scale
    base = recipe_tensor(tag + ".base", w.shape).to(dev)
    acc = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), Cout, Cin, k, stride, scale=0.5, out=base.clone(), accumulate=True)
    assert rel_l2(acc, base.cpu() + 0.5 * w.grad) < TOL


def test_wgrad_with_upsampled_and_bn_folded_input(pkg, dev):
    B, Cin, Cout, Hs = 2, 24, 40, 10
    x = recipe_input("wgu.x", (B, Cin, Hs, Hs))
    w = recipe_tensor("wgu.weight", (Cout, Cin, 3, 3)).requires_grad_(True)
    y = F.conv2d(R.upsample2x_bilinear(x), w, padding=1)
    g = recipe_input("wgu.g", y.shape)
    y.backward(g)
    assert rel_l2(pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), Cout, Cin, 3, 1, upsample=True), w.grad) < TOL
    a, b = 1.0 + recipe_tensor("wgu.a", (Cin,), 0.3), recipe_tensor("wgu.b", (Cin,), 0.3)
    w2 = recipe_tensor("wgu.w2", (Cout, Cin, 3, 3)).requires_grad_(True)
    y2 = F.conv2d(F.relu(x * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)), w2, stride=2, padding=1)
    g2 = recipe_input("wgu.g2", y2.shape)
    y2.backward(g2)
    dw = pkg.ops.conv2d_wgrad(g2.to(dev), x.to(dev), Cout, Cin, 3, 2, in_affine=(a.to(dev), b.to(dev)))
    assert rel_l2(dw, w2.grad) < TOL


@pytest.mark.parametrize("B,Cin,Cout,H,W,groups,aff", [
    (2, 64, 64, 8, 8, 1, False),         # 8 x 8 tiles: one per image (the 8^2 layers)
    (3, 70, 130, 12, 12, 1, True),       # 8 x 8 tiles, partial in both directions, ragged channel blocks, folded BatchNorm
    (2, 64, 64, 8, 8, 3, True),          # grouped (the trunk's layer4 convs)
    (2, 40, 72, 6, 8, 1, False),
    (2, 64, 128, 20, 16, 2, True),       # 16 x 4 tiles, grouped, folded BatchNorm
])
def test_wgrad_stride1_wide_form(pkg, dev, B, Cin, Cout, H, W, groups, aff):
    """wgrad3x3_wide_kernel<16 | 8> (64co x 64ci blocks, 16-byte row loads, staging behind the MFMAs) against autograd."""
    tag = f"wgw.{B}.{Cin}.{Cout}.{H}.{W}.{groups}.{int(aff)}"
    G = groups
    x = recipe_input(tag + ".x", (B, G * Cin, H, W))
    a = 1.0 + recipe_tensor(tag + ".a", (G * Cin,), 0.3)
    b = recipe_tensor(tag + ".b", (G * Cin,), 0.3)
    xin = F.relu(x * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)) if aff else x
    ws = [recipe_tensor(tag + f".weight{q}", (Cout, Cin, 3, 3)).requires_grad_(True) for q in range(G)]
    y = torch.cat([F.conv2d(xin[:, q * Cin:(q + 1) * Cin], ws[q], padding=1) for q in range(G)], 1)
    g = recipe_input(tag + ".g", y.shape)
    y.backward(g)
    ref = torch.cat([w.grad for w in ws], 0)
    kw = dict(in_affine=(a.to(dev), b.to(dev))) if aff else {}
    for splits in (0, 1, 3):
        dw = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), Cout, Cin, 3, 1, splits=splits, groups=G, **kw)
        assert rel_l2(dw, ref) < TOL, splits
    base = recipe_tensor(tag + ".base", ref.shape).to(dev)
    acc = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), Cout, Cin, 3, 1, scale=0.5, out=base.clone(), accumulate=True, groups=G, **kw)
    assert rel_l2(acc, base.cpu() + 0.5 * ref) < TOL


@pytest.mark.parametrize("B,Hin,Win,G,shared", [(2, 40, 40, 1, False), (3, 50, 72, 2, False), (2, 34, 136, 4, True), (1, 128, 128, 2, True)])
def test_wgrad_stem_form(pkg, dev, B, Hin, Win, G, shared):
    """wgrad_stem_kernel (7x7 stride 2, 3 -> 64 per group: 64 x 147 GEMM over the pixels, persistent workgroups, one slab each)
    against autograd: partial tiles, grouped with own / shared images, slab counts, fold, scale + accumulate."""
    tag = f"wgst.{B}.{Hin}.{Win}.{G}.{int(shared)}"
    x = recipe_input(tag + ".x", (B, 3 if shared else 3 * G, Hin, Win), "uniform")
    ws = [recipe_tensor(tag + f".weight{q}", (64, 3, 7, 7)).requires_grad_(True) for q in range(G)]
    y = torch.cat([F.conv2d(x if shared else x[:, 3 * q:3 * q + 3], ws[q], stride=2, padding=3) for q in range(G)], 1)
    g = recipe_input(tag + ".g", y.shape)
    y.backward(g)
    ref = torch.cat([w.grad for w in ws], 0)
    for splits in (0, 1, 7):
        dw = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), 64, 3, 7, 2, splits=splits, groups=G, shared_input=shared)
        assert rel_l2(dw, ref) < TOL, splits
    if G % 2 == 0:
        dw = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), 64, 3, 7, 2, groups=G, shared_input=shared, fold=2)
        assert rel_l2(dw, ref[:G // 2 * 64] + ref[G // 2 * 64:]) < TOL
    base = recipe_tensor(tag + ".base", ref.shape).to(dev)
    acc = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), 64, 3, 7, 2, scale=0.5, out=base.clone(), accumulate=True, groups=G, shared_input=shared)
    assert rel_l2(acc, base.cpu() + 0.5 * ref) < TOL


@pytest.mark.parametrize("B,Cin,Cout,H,W,groups,aff", [
    (2, 64, 64, 8, 8, 1, False),         # 64 x 64 block (one MFMA tile per wave), two k-tiles per image
    (3, 128, 64, 8, 4, 2, True),         # 64 co x 128 ci, grouped, folded BatchNorm, an odd number of k-tiles (3)
    (2, 64, 128, 16, 16, 1, True),       # 128 co x 64 ci
    (2, 128, 256, 8, 8, 3, False),       # 128 x 128 blocks, two co blocks per group, grouped
    (1, 256, 128, 4, 8, 1, True),        # ONE k-tile in all: the ring's prologue alone
    (5, 128, 128, 8, 12, 2, True),       # k-tiles that cross images (96 pixels per image = 3 tiles), folded pair of groups below
])
def test_wgrad_1x1_lds_dma_form(pkg, dev, B, Cin, Cout, H, W, groups, aff):
    """wgrad1x1_dma_kernel (whole 64 / 128-channel blocks, H W % 32 == 0: operands by LDS-DMA into XOR-swizzled 128-byte rows,
    ds_read_b128 fragments, two ring slots) against autograd: split counts, fold, scale + accumulate."""
    tag = f"wgd.{B}.{Cin}.{Cout}.{H}.{W}.{groups}.{int(aff)}"
    G = groups
    x = recipe_input(tag + ".x", (B, G * Cin, H, W))
    a = 1.0 + recipe_tensor(tag + ".a", (G * Cin,), 0.3)
    b = recipe_tensor(tag + ".b", (G * Cin,), 0.3)
    xin = F.relu(x * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)) if aff else x
    ws = [recipe_tensor(tag + f".weight{q}", (Cout, Cin, 1, 1)).requires_grad_(True) for q in range(G)]
    y = torch.cat([F.conv2d(xin[:, q * Cin:(q + 1) * Cin], ws[q]) for q in range(G)], 1)
    g = recipe_input(tag + ".g", y.shape)
    y.backward(g)
    ref = torch.cat([w.grad for w in ws], 0)
    kw = dict(in_affine=(a.to(dev), b.to(dev))) if aff else {}
    for splits in (0, 1, 2, 5):
        dw = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), Cout, Cin, 1, 1, splits=splits, groups=G, **kw)
        assert rel_l2(dw, ref) < TOL, splits
    if G % 2 == 0:
        dw = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), Cout, Cin, 1, 1, groups=G, fold=2, **kw)
        assert rel_l2(dw, ref[:G // 2 * Cout] + ref[G // 2 * Cout:]) < TOL
    base = recipe_tensor(tag + ".base", ref.shape).to(dev)
    acc = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), Cout, Cin, 1, 1, scale=0.5, out=base.clone(), accumulate=True, groups=G, **kw)
    assert rel_l2(acc, base.cpu() + 0.5 * ref) < TOL


@pytest.mark.parametrize("B,Cin,Cout,H,W,groups,aff", [
    (2, 32, 128, 16, 16, 1, False),      # 16 x 4 output tiles, whole blocks
    (3, 40, 160, 10, 20, 1, True),       # ragged channels (8 live rows in the second ci block, 32 in the second co block), partial tiles
    (2, 64, 128, 8, 8, 1, False),        # 8 x 8 tiles (the 16^2 -> 8^2 layers)
    (3, 24, 96, 6, 8, 1, True),          # 8 x 8 tiles, partial in y, fewer channels than a block
    (2, 32, 128, 12, 16, 3, True),       # grouped (the three encoders), BatchNorm-folded input
    (2, 32, 256, 8, 8, 2, False),
])
def test_wgrad_stride2_wide_form(pkg, dev, B, Cin, Cout, H, W, groups, aff):
    """wgrad3x3_s2_kernel (128co x 32ci blocks, 16-byte row loads, staging behind the MFMAs) against autograd; H, W = OUTPUT size."""
    tag = f"wgs2.{B}.{Cin}.{Cout}.{H}.{W}.{groups}.{int(aff)}"
    G = groups
    x = recipe_input(tag + ".x", (B, G * Cin, 2 * H, 2 * W))
    a = 1.0 + recipe_tensor(tag + ".a", (G * Cin,), 0.3)
    b = recipe_tensor(tag + ".b", (G * Cin,), 0.3)
    xin = F.relu(x * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)) if aff else x
    ws = [recipe_tensor(tag + f".weight{q}", (Cout, Cin, 3, 3)).requires_grad_(True) for q in range(G)]
    y = torch.cat([F.conv2d(xin[:, q * Cin:(q + 1) * Cin], ws[q], stride=2, padding=1) for q in range(G)], 1)
    g = recipe_input(tag + ".g", y.shape)
    y.backward(g)
    ref = torch.cat([w.grad for w in ws], 0)
    kw = dict(in_affine=(a.to(dev), b.to(dev))) if aff else {}
    for splits in (0, 1, 5):
        dw = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), Cout, Cin, 3, 2, splits=splits, groups=G, **kw)
        assert rel_l2(dw, ref) < TOL, splits
    if G > 1:     # the two images' gradients folded in the slab reduce
        if G % 2 == 0:
            dw = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), Cout, Cin, 3, 2, groups=G, fold=2, **kw)
            assert rel_l2(dw, ref[:Cout] + ref[Cout:]) < TOL
    base = recipe_tensor(tag + ".base", ref.shape).to(dev)
    acc = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), Cout, Cin, 3, 2, scale=0.5, out=base.clone(), accumulate=True, groups=G, **kw)
    assert rel_l2(acc, base.cpu() + 0.5 * ref) < TOL


@pytest.mark.parametrize("B,I,O", [(5, 512, 96), (16, 512, 512), (9, 512, 1024), (8, 6144, 512), (3, 2048, 64), (16, 48, 512), (20, 64, 1024)])
def test_fc_shape_specialised_kernels(pkg, dev, B, I, O):
    """The FC forms picked by shape -- 512-wide rows with every load in flight (fc_dot512), four rows per workgroup for long rows
    (fc_wide4_kernel, I = 1024 U), the input gradient on 16-column slabs (fc_bwd_input_cols_kernel, O % 512 == 0) and the general
    kernels -- against torch, forward and backward, more rows than one batch tile."""
    tag = f"fcs.{B}.{I}.{O}"
    x = recipe_input(tag + ".x", (B, I)).requires_grad_(True)
    w = recipe_tensor(tag + ".w", (O, I)).requires_grad_(True)
    b = recipe_tensor(tag + ".b", (O,)).requires_grad_(True)
    wmul, bmul, slope = 0.37, 1.3, 0.2
    ref = F.leaky_relu(F.linear(x, w * wmul, b * bmul), slope)
    g = recipe_input(tag + ".g", ref.shape)
    ref.backward(g)
    out = pkg.ops.fc(x.detach().to(dev), w.detach().to(dev), b.detach().to(dev), wmul, bmul, slope)
    assert rel_l2(out, ref.detach()) < TOL
    dx, dw, db = pkg.ops.fc_bwd(g.to(dev), out, x.detach().to(dev), w.detach().to(dev), wmul, bmul, slope)
    assert rel_l2(dx, x.grad) < TOL and rel_l2(dw, w.grad) < TOL and rel_l2(db, b.grad) < TOL


def test_pointwise_backward_kernels(pkg, dev, golden):
    # bilinear x2 adjoint, incl. odd sizes and 1x1
    g = golden("decoder_ops.npz")
    for tag in ("up_a", "up_b", "up_c"):
        assert rel_l2(pkg.ops.upsample2x_bilinear_bwd(T(g[f"{tag}.gy"]).to(dev)), g[f"{tag}.gx"]) < TOL
    # widths that are multiples of 4 take the four-pixels-per-thread kernel (16-byte loads): borders, one-row planes, many planes
    for i, shp in enumerate([(2, 3, 4, 4), (1, 2, 8, 12), (2, 2, 16, 4), (1, 1, 1, 4), (3, 5, 32, 32)]):
        xs = recipe_input(f"upv{i}.x", shp).requires_grad_(True)
        ys = F.interpolate(xs, scale_factor=2, mode="bilinear", align_corners=False)
        gy = recipe_input(f"upv{i}.g", ys.shape)
        ys.backward(gy)
        assert rel_l2(pkg.ops.upsample2x_bilinear_bwd(gy.to(dev)), xs.grad) < TOL, shp
    # fused-epilogue adjoint vs autograd of the op chain
    B, C, H = 3, 10, 7
    t = recipe_input("eb.t", (B, C, H, H)).requires_grad_(True)
    nz = recipe_input("eb.nz", (B, 1, H, H))
    nw = recipe_tensor("eb.nw", (C,), 0.5).requires_grad_(True)
    st = recipe_input("eb.st", (B, 2 * C)).requires_grad_(True)
    a = F.leaky_relu(t + nw.view(1, -1, 1, 1) * nz, 0.2)
    y = a * (st[:, :C].view(B, C, 1, 1) + 1.0) + st[:, C:].view(B, C, 1, 1)
    dy = recipe_input("eb.dy", y.shape)
    y.backward(dy)
    dt, sums = pkg.ops.epilogue_bwd(dy.to(dev), a.detach().to(dev), nz.to(dev), st.detach().to(dev), 0.2)
    assert rel_l2(dt, t.grad) < TOL
    assert rel_l2(sums[:, :2].reshape(sums.shape[0], -1), st.grad) < TOL      # [d s0 | d s1], a view of rows 0-1
    assert rel_l2(sums[:, 3].sum(0), nw.grad) < TOL
    assert rel_l2(sums[:, 2].sum(0), t.grad.sum((0, 2, 3))) < TOL
    # toRGB backward
    x = recipe_input("rgbb.x", (2, 16, 12, 12)).requires_grad_(True)
    w = recipe_tensor("rgbb.w", (3, 16, 1, 1)).requires_grad_(True)
    b = recipe_tensor("rgbb.b", (3,)).requires_grad_(True)
    y = F.conv2d(x, w, b)
    dy = recipe_input("rgbb.dy", y.shape)
    y.backward(dy)
    dx, dw, db = pkg.ops.conv1x1_small_bwd(x.detach().to(dev), w.detach().to(dev), dy.to(dev))
    assert rel_l2(dx, x.grad) < TOL and rel_l2(dw, w.grad) < TOL and rel_l2(db, b.grad) < TOL


def test_fc_backward_goldens(pkg, dev, golden):
    g = golden("decoder_ops.npz")
    for tag, gain, wscale, lrmul, has_bias in [("fc_map", 2 ** 0.5, True, 0.01, True), ("fc_style", 1.0, True, 1.0, True),
                                               ("fc_plain", 2 ** 0.5, False, 1.0, False)]:
        O, I = g[f"{tag}.gw"].shape
        m = pkg.FC(I, O, gain=gain, use_wscale=wscale, lrmul=lrmul, bias=has_bias)
        m.load_state_dict(fill_state_dict(m.state_dict(), prefix=tag + "."))
        m.to(dev)
        x = T(g[f"{tag}.x"]).to(dev).requires_grad_(True)
        m(x).backward(T(g[f"{tag}.gy"]).to(dev))
        assert rel_l2(x.grad, g[f"{tag}.gx"]) < TOL, tag
        assert rel_l2(m.weight.grad, g[f"{tag}.gw"]) < TOL, tag
        if has_bias:
            assert rel_l2(m.bias.grad, g[f"{tag}.gb"]) < TOL, tag
    # stand-alone ApplyNoise / ApplyStyle modules
    C = g["an.x"].shape[1]
    m = pkg.ApplyNoise(C)
    m.load_state_dict(fill_state_dict(m.state_dict(), prefix="an."))
    m.to(dev)
    x = T(g["an.x"]).to(dev).requires_grad_(True)
    m(x, T(g["an.noise"]).to(dev)).backward(T(g["an.gy"]).to(dev))
    assert rel_l2(x.grad, g["an.gx"]) < TOL and rel_l2(m.weight.grad, g["an.gw"]) < TOL
    m = pkg.ApplyStyle(16, C, use_wscale=True)
    m.load_state_dict(fill_state_dict(m.state_dict(), prefix="as."))
    m.to(dev)
    x, lat = T(g["as.x"]).to(dev).requires_grad_(True), T(g["as.lat"]).to(dev).requires_grad_(True)
    m(x, lat).backward(T(g["as.gy"]).to(dev))
    for got, k in [(x.grad, "as.gx"), (lat.grad, "as.glat"), (m.linear.weight.grad, "as.gw"), (m.linear.bias.grad, "as.gb")]:
        assert rel_l2(got, g[k]) < TOL, k


def test_synthesis_block_gradients_vs_reference_goldens(pkg, dev, golden):
    g = golden("decoder_blocks.npz")
    for tag, cin, cout, B, hin in [("blk512", 512, 512, 2, 4), ("blk128_64", 128, 64, 1, 16), ("blk16_8", 16, 8, 3, 6)]:
        m = pkg.SynthesisBlock(cin, cout, 3)
        m.load_state_dict(fill_state_dict(m.state_dict(), prefix=tag + "."))
        m.to(dev)
        x = recipe_input(tag + ".x", (B, cin, hin, hin)).to(dev).requires_grad_(True)
        w = recipe_input(tag + ".w", (B, 2, 512)).to(dev).requires_grad_(True)
        n1 = recipe_input(tag + ".n1", (B, 1, 2 * hin, 2 * hin)).to(dev)
        n2 = recipe_input(tag + ".n2", (B, 1, 2 * hin, 2 * hin)).to(dev)
        y = m(x, w, n1, n2)
        assert rel_l2(y, g[f"{tag}.y"]) < TOL
        y.backward(recipe_input(tag + ".gy", y.shape).to(dev))
        assert rel_l2(x.grad, g[f"{tag}.gx"]) < 1e-4, tag
        assert rel_l2(w.grad, g[f"{tag}.gw"]) < 1e-4, tag
        for pn, p in m.named_parameters():
            if f"{tag}.g.{pn}" in g:
                assert rel_l2(p.grad, g[f"{tag}.g.{pn}"]) < 1e-4, (tag, pn)
            else:
                sl = p.grad[:8, :8] if p.grad.dim() == 4 else p.grad[:8, :64]
                assert rel_l2(sl, g[f"{tag}.g.{pn}.slice"]) < 1e-4, (tag, pn)
                assert abs(float(p.grad.double().norm()) / float(g[f"{tag}.g.{pn}.norm"]) - 1) < 1e-4, (tag, pn)


def test_style_generator_train_step_vs_reference_golden(pkg, dev, golden):
    """Golden G4: train mode, style mixing taken, full backward from a recipe output gradient."""
    g = golden("decoder_train_mix.npz")
    gen = pkg.StyleGenerator(6144).train()
    gen.load_state_dict(fill_state_dict(gen.state_dict(), prefix="Gd."))
    gen.to(dev)
    feats = recipe_input("mix.features", (1, 6144)).to(dev).requires_grad_(True)
    y = gen(feats, [n.to(dev) for n in recipe_noises("mix", 1, 256)],
            style_mix=(T(g["mix_features"]).to(dev), int(g["mix_layer"])))
    assert rel_l2(y[..., ::4, ::4], g["y_s4"]) < 1e-4
    y.backward(recipe_input("mix.gy", y.shape).to(dev))
    s = gen.synthesis
    checks = [(feats.grad, "gfeat"), (gen.mapping[7].bias.grad, "g_map7_bias"), (s.const_input.grad, "g_const"),
              (s.to_rgb.weight.grad, "g_rgb_w"), (s.to_rgb.bias.grad, "g_rgb_b"), (s.layers[5].noise2.weight.grad, "g_l5_noise2"),
              (s.layers[5].conv2.bias.grad, "g_l5_conv2_b"), (s.layers[5].conv2.weight.grad, "g_l5_conv2_w"),
              (s.layers[0].conv1.weight.grad[:8, :8], "g_l0_conv1_w_slice"), (s.layers[3].style_mod1.linear.bias.grad, "g_l3_style1_b")]
    # The reference's own fp32 gradients are only ~1e-3 accurate here (random-sign output gradient ->
    # cancelling sums over 65,536 pixels; measured: its fp32 values sit 1.2e-3..1.7e-3 rel-L2 from an
    # fp64 evaluation of the same graph).  So: (1) stay within 3e-3 of the reference's numbers, and
    # (2) stay within 3e-3 of the fp64 truth as well.  (The noise is LeakyReLU-mask flips: an activation
    # within ~1e-6 of zero gets the other slope in one implementation; with a random-sign output gradient
    # a single flipped pixel moves a 65,536-term cancelling sum by ~4e-3 of its value.)
    sd64 = {k: v.double().requires_grad_(True) for k, v in fill_state_dict(gen.state_dict(), prefix="Gd.").items()}
    f64 = recipe_input("mix.features", (1, 6144)).double().requires_grad_(True)
    y64 = R.style_generator(f64, sd64, [n.double() for n in recipe_noises("mix", 1, 256)],
                            mix_features=T(g["mix_features"]).double(), mix_layer=int(g["mix_layer"]))
    y64.backward(recipe_input("mix.gy", y64.shape).double())
    truth = {"gfeat": f64.grad, "g_map7_bias": sd64["mapping.7.bias"].grad, "g_const": sd64["synthesis.const_input"].grad,
             "g_rgb_w": sd64["synthesis.to_rgb.weight"].grad, "g_rgb_b": sd64["synthesis.to_rgb.bias"].grad,
             "g_l5_noise2": sd64["synthesis.layers.5.noise2.weight"].grad,
             "g_l5_conv2_b": sd64["synthesis.layers.5.conv2.bias"].grad,
             "g_l5_conv2_w": sd64["synthesis.layers.5.conv2.weight"].grad,
             "g_l0_conv1_w_slice": sd64["synthesis.layers.0.conv1.weight"].grad[:8, :8],
             "g_l3_style1_b": sd64["synthesis.layers.3.style_mod1.linear.bias"].grad}
    for got, k in checks:
        e_ref, e_truth, ref_truth = rel_l2(got, g[k]), rel_l2(got, truth[k]), rel_l2(T(g[k]), truth[k])
        assert e_ref < 3e-3, (k, e_ref)
        assert e_truth < 3e-3 and ref_truth < 3e-3, (k, e_truth, ref_truth)
    assert abs(float(s.layers[0].conv1.weight.grad.double().norm()) / float(g["g_l0_conv1_w_norm"]) - 1) < 3e-3


def test_batch8_training_step_vs_oracle(pkg, dev):
    """B=8 256^2 (the benchmarked shape) fwd+bwd of mean((G(z) - target)^2) against the CPU oracle."""
    gen = pkg.StyleGenerator(6144).train()
    sd = fill_state_dict(gen.state_dict(), prefix="Gd.")
    gen.load_state_dict(sd)
    gen.to(dev)
    B = 4
    feats = recipe_input("ts.features", (B, 6144))
    noises = recipe_noises("ts", B, 256)
    target = recipe_input("ts.target", (B, 3, 256, 256), "uniform")
    y = gen(feats.to(dev), [n.to(dev) for n in noises], style_mix=False)
    loss = ((y - target.to(dev)) ** 2).mean()
    loss.backward()
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y_ref = R.style_generator(feats, sd_ref, noises)
    loss_ref = ((y_ref - target) ** 2).mean()
    loss_ref.backward()
    assert abs(loss.item() / loss_ref.item() - 1) < 1e-4
    got = dict(gen.named_parameters())
    # noise-weight gradients are cancelling sums of dt*noise over every pixel: the most exposed to the
    # LeakyReLU-mask flips described above -> 2e-3; everything else 5e-4
    tol = lambda k: 2e-3 if "noise" in k else 5e-4
    for k in ("mapping.0.weight", "mapping.7.bias", "synthesis.const_input", "synthesis.bias", "synthesis.noise_input1.weight",
              "synthesis.style_mod.linear.weight", "synthesis.layers.0.conv1.weight", "synthesis.layers.2.conv2.weight",
              "synthesis.layers.3.conv1.bias", "synthesis.layers.4.noise1.weight", "synthesis.layers.5.conv1.weight",
              "synthesis.layers.5.style_mod2.linear.weight", "synthesis.to_rgb.weight"):
        assert rel_l2(got[k].grad, sd_ref[k].grad) < tol(k), k


@pytest.mark.parametrize("B,Cin,Cout,Hs,Ws", [(2, 64, 64, 16, 16), (1, 20, 40, 12, 12), (3, 70, 130, 6, 20), (2, 64, 64, 32, 32),
                                               (1, 128, 72, 10, 36), (8, 128, 64, 16, 16)])
def test_wgrad_of_upsampled_input_without_materialising_it(pkg, dev, B, Cin, Cout, Hs, Ws):
    """dW of conv3x3(bilinear_x2(x)) from the LOW-resolution x (SPK_CONV_UPSAMPLE2X in spk_conv2d_wgrad: the x2 plane is
    interpolated LDS -> LDS from a source patch; round 1 wrote the x2 tensor to HBM first) against autograd of
    F.interpolate + F.conv2d -- partial tiles in both directions, ragged channel blocks, one workgroup walking many tiles
    (splits = 1) so that the two-tile-ahead pipeline is exercised, and the accumulate / scale form."""
    tag = f"wgu.{B}.{Cin}.{Cout}.{Hs}.{Ws}"
    x = recipe_input(tag + ".x", (B, Cin, Hs, Ws))
    w = recipe_tensor(tag + ".weight", (Cout, Cin, 3, 3)).requires_grad_(True)
    y = F.conv2d(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False), w, padding=1)
    g = recipe_input(tag + ".g", y.shape)
    y.backward(g)
    # (tests/conftest.py sets SPK_WGRAD_UP_MIN_W=16: in the test process every shape the folded kernel can take goes to it)
    assert pkg._lib.lib().spk_conv2d_wgrad_up_supported(B, Cin, Cout, 2 * Hs, 2 * Ws)
    for splits in (0, 1, 3):
        dw = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), Cout, Cin, 3, 1, upsample=True, splits=splits)
        assert rel_l2(dw, w.grad) < TOL, splits
    base = recipe_tensor(tag + ".base", w.shape).to(dev)
    acc = pkg.ops.conv2d_wgrad(g.to(dev), x.to(dev), Cout, Cin, 3, 1, upsample=True, scale=0.5, out=base.clone(), accumulate=True)
    assert rel_l2(acc, base.cpu() + 0.5 * w.grad) < TOL


def test_decoder_gradients_with_the_default_upsample_threshold():
    """conftest.py forces SPK_WGRAD_UP_MIN_W=16 for this process, so the shipped dispatch of the 16 <= W < 128 upsample
    layers (materialise the x2 image, plain weight-gradient kernel) would never meet the decoder's block / end-to-end
    gradient goldens.  The library reads the variable once, so those tests run again in a child process without it."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "SPK_WGRAD_UP_MIN_W"}
    env["SPK_WGRAD_UP_MIN_W"] = "128"                       # the product default (csrc/wgrad_mfma_f32.hip), stated
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "synthesis_block_gradients or style_generator_train_step or batch8_training_step"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
