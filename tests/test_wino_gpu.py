"""The fp32 Winograd F(2x2, 3x3) conv (csrc/conv3x3_wino_f32.hip, SPK_CONV_WINOGRAD) against an fp64 convolution of the same
operands and against the direct f32 MFMA kernel, through the C ABI.  It replaces the F.conv2d of styleganv1.py:625,630,662
where the shape allows.  Tolerance: 5e-6 rel-L2 per layer against fp64 (the direct kernel sits at ~1e-7: an fmaf chain; the
Winograd form adds the rounding of its +-1 / 0.5 transforms -- the path's bound, north_star, is 1e-3)."""
import importlib
import os
import sys

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TOL = 5e-6


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    return importlib.import_module("speak-hack_amd").ops


def _ref(x, w, **kw):
    y = F.conv2d(x.double(), w.double(), padding=1)
    if kw.get("out_scale") is not None:
        y = y * kw["out_scale"]
    if kw.get("bias") is not None:
        y = y + kw["bias"].double().view(1, -1, 1, 1)
    if kw.get("noise") is not None:
        y = y + kw["noise_w"].double().view(1, -1, 1, 1) * kw["noise"].double()
    if kw.get("slope") is not None:
        y = F.leaky_relu(y, kw["slope"])
    pre = y
    if kw.get("style") is not None:
        C = w.shape[0]
        s = kw["style"].double()
        y = y * (s[:, :C].view(-1, C, 1, 1) + 1) + s[:, C:].view(-1, C, 1, 1)
    return y, pre


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 16, 64, 8, 32), (1, 32, 48, 16, 32), (2, 64, 64, 32, 64), (1, 512, 512, 32, 32),
                                            (3, 128, 64, 24, 96), (1, 16, 200, 8, 64)])
def test_wino_plain_equals_fp64_conv(ops, B, Cin, Cout, H, W):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
    assert ops.wino_supported(B, Cin, Cout, H, W)
    y = ops.conv3x3_wino(x, ops.pack_conv_weight_wino(w), Cout)
    ref, _ = _ref(x, w)
    assert rel_l2(y, ref) < TOL, rel_l2(y, ref)
    cfg = ops.conv2d_pick_config(3, 1, B, Cin, Cout, H, W)
    direct = ops.conv2d_fused(x, ops.pack_conv_weight(w, cfg), Cout, 3, 1, config=cfg)
    assert rel_l2(y, direct) < TOL
    # every border pixel of every region, not only the aggregate: the zero padding comes from out-of-range gathers
    err = (y.double() - ref).abs().amax(dim=(0, 1))
    assert float(err.max()) < 1e-4 * float(ref.abs().max()), float(err.max())


def test_wino_full_epilogue_and_pre_style_output(ops):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    B, Cin, Cout, H, W = 2, 64, 128, 16, 64
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
    bias, nw = torch.randn(Cout, generator=g).to(dev), torch.randn(Cout, generator=g).to(dev)
    noise = torch.randn(B, 1, H, W, generator=g).to(dev)
    style = torch.randn(B, 2 * Cout, generator=g).to(dev)
    pre = torch.empty(B, Cout, H, W, device=dev)
    y = ops.conv3x3_wino(x, ops.pack_conv_weight_wino(w), Cout, bias=bias, noise_w=nw, noise=noise, style=style, lrelu_slope=0.2,
                         out_scale=0.7, out_pre=pre)
    ref, ref_pre = _ref(x, w, bias=bias, noise_w=nw, noise=noise, style=style, slope=0.2, out_scale=0.7)
    assert rel_l2(y, ref) < TOL and rel_l2(pre, ref_pre) < TOL
    # the same through the direct kernel's epilogue: the two agree element for element up to the contraction's rounding
    cfg = ops.conv2d_pick_config(3, 1, B, Cin, Cout, H, W)
    d = ops.conv2d_fused(x, ops.pack_conv_weight(w, cfg), Cout, 3, 1, bias=bias, noise_w=nw, noise=noise, style=style, lrelu_slope=0.2,
                         out_scale=0.7, config=cfg)
    assert rel_l2(y, d) < TOL


def test_wino_accumulate_device_scale_and_data_gradient(ops):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    B, Cin, Cout, H, W = 2, 32, 64, 16, 32
    x = torch.randn(B, Cin, H, W, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
    base = torch.randn(B, Cout, H, W, generator=g).to(dev)
    sd = torch.tensor([0.37], device=dev)
    y = ops.conv3x3_wino(x.detach(), ops.pack_conv_weight_wino(w), Cout, out=base.clone(), accumulate=True, out_scale=2.0, out_scale_dev=sd)
    ref = base.double() + 2.0 * 0.37 * F.conv2d(x.detach().double(), w.double(), padding=1)
    assert rel_l2(y, ref) < TOL
    # data gradient = the same kernel on the transposed, flipped weights (Cin / Cout exchanged)
    dy = torch.randn(B, Cout, H, W, generator=g).to(dev)
    (dx_ref,) = torch.autograd.grad(F.conv2d(x.double(), w.double(), padding=1), x, dy.double())
    dx = ops.conv3x3_wino(dy, ops.pack_conv_weight_wino(w, transpose_flip=True), Cin)
    assert rel_l2(dx, dx_ref) < TOL


def test_wino_rejects_what_it_does_not_serve(ops):
    assert not ops.wino_supported(1, 24, 64, 8, 32)      # an odd number of 8-channel chunks
    assert not ops.wino_supported(1, 16, 64, 12, 32)     # not whole 32 x 8 (or 16 x 16) regions
    assert not ops.wino_supported(1, 16, 64, 8, 16)
    assert ops.wino_supported(1, 16, 64, 16, 16) and ops.wino_supported(1, 16, 64, 32, 48)      # 16 x 16 regions
    L = importlib.import_module("speak-hack_amd")._lib
    dev = torch.device("cuda:0")
    x, w = torch.randn(1, 16, 12, 32, device=dev), torch.randn(64, 16, 3, 3, device=dev)
    with pytest.raises(L.SpkError):
        ops.conv3x3_wino(x, ops.pack_conv_weight_wino(w), 64)


@pytest.mark.parametrize("shape", [(3, 5, 4, 4), (2, 7, 8, 16), (1, 130, 64, 64), (2, 3, 6, 10), (1, 2, 1, 4), (1, 1, 5, 1)])
def test_upsample2x_launch_equals_torch_bilinear(ops, shape):
    """spk_upsample2x_bilinear_fwd (nn.Upsample(scale_factor=2, mode='bilinear'), styleganv1.py:621): the 16-byte-access form taken
    when the source width is a multiple of 4 -- it writes the x2 image a Winograd x2 layer reads -- and the generic form otherwise."""
    dev = torch.device("cuda:0")
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape))).to(dev)
    y = ops.upsample2x_bilinear(x)
    ref = F.interpolate(x.double(), scale_factor=2, mode="bilinear", align_corners=False)
    assert y.shape == ref.shape
    assert float((y.double() - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))


def test_wino_modulated_conv_equals_fp64(ops):
    """The StyleGAN2 variant's styled conv on the Winograd kernel (SPK_CONV_WINOGRAD | SPK_CONV_IN_BATCH_SCALE): y = gain * lrelu(d[b,co]
    * scale * conv3x3(x * s[b,ci], w) + noise_w * noise + bias), the modulation applied to the transformed input, the demodulation in
    the epilogue -- against the formula in fp64 and against the direct kernel's modulated form."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(23)
    B, Cin, Cout, H, W = 3, 64, 96, 16, 64
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
    s = (torch.rand(B, Cin, generator=g) + 0.5).to(dev)
    d = (torch.rand(B, Cout, generator=g) + 0.5).to(dev)
    bias, nw = torch.randn(Cout, generator=g).to(dev), torch.randn(Cout, generator=g).to(dev)
    noise = torch.randn(B, 1, H, W, generator=g).to(dev)
    y = ops.conv3x3_wino(x, ops.pack_conv_weight_wino(w), Cout, bias=bias, noise_w=nw, noise=noise, lrelu_slope=0.2, out_scale=0.3,
                         batch_scale=s, demod=d, act_gain=2 ** 0.5)
    c = F.conv2d(x.double() * s.double().view(B, Cin, 1, 1), w.double(), padding=1) * 0.3 * d.double().view(B, Cout, 1, 1)
    ref = F.leaky_relu(c + bias.double().view(1, -1, 1, 1) + nw.double().view(1, -1, 1, 1) * noise.double(), 0.2) * 2 ** 0.5
    assert rel_l2(y, ref) < TOL, rel_l2(y, ref)
    cfg = ops.conv2d_pick_config(3, 1, B, Cin, Cout, H, W)
    cfg = cfg + 4 if cfg < 4 else cfg
    direct = ops.conv2d_fused(x, ops.pack_conv_weight(w, cfg), Cout, 3, 1, bias=bias, noise_w=nw, noise=noise, lrelu_slope=0.2, out_scale=0.3,
                              batch_scale=s, demod=d, act_gain=2 ** 0.5, config=cfg)
    assert rel_l2(y, direct) < TOL


@pytest.mark.parametrize("shape", [(2, 5, 4, 8), (1, 3, 16, 32)])
def test_upsample2x_zero_border_equals_upfirdn2d(ops, shape):
    """spk_upsample2x_fwd(zero_border = 1) = upfirdn2d(up = 2, [1,3,3,1] * 4 / 64, pad (2,1)): what a modulated Winograd x2 layer reads."""
    dev = torch.device("cuda:0")
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(5)).to(dev)
    y = ops.upsample2x(x, zero_border=True)
    k1 = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=torch.float64)
    k = (k1[:, None] * k1[None, :]) / 64.0 * 4.0
    C = shape[1]
    up = torch.zeros(shape[0], C, 2 * shape[2], 2 * shape[3], dtype=torch.float64)
    up[:, :, ::2, ::2] = x.double().cpu()
    ref = F.conv2d(F.pad(up, (2, 1, 2, 1)), k.flip(0, 1).view(1, 1, 4, 4).repeat(C, 1, 1, 1), groups=C)
    assert float((y.double().cpu() - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))


# ---- the weight gradient as Winograd F(2x2, 3x3) (csrc/wgrad3x3_wino_f32.hip) ------------------------------------------------
def _wgrad_ref(g, x):
    w = torch.zeros(g.shape[1], x.shape[1], 3, 3, device=g.device, dtype=torch.float64, requires_grad=True)
    return torch.autograd.grad(F.conv2d(x.double(), w, padding=1), w, g.double())[0]


@pytest.mark.parametrize("B,Cin,Cout,H,W,splits", [(1, 64, 64, 8, 32, 1), (1, 64, 64, 8, 32, 2), (2, 64, 64, 4, 16, 0), (1, 128, 64, 8, 64, 1),
                                                   (4, 64, 128, 16, 32, 0), (3, 64, 64, 6, 48, 0), (2, 192, 128, 32, 32, 0),
                                                   (1, 64, 64, 2, 64, 0)])
def test_wgrad_wino_equals_fp64_weight_gradient(ops, B, Cin, Cout, H, W, splits):
    """Every image border, every chunk boundary between workgroups (splits), ragged chunk counts, and twice: the result is a
    fixed-order sum (bitwise reproducible)."""
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(B * 100 + Cin + H + W + splits)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dev)
    g = torch.randn(B, Cout, H, W, generator=gen).to(dev)
    assert ops.wgrad_wino_supported(B, Cin, Cout, H, W)
    dw = ops.conv2d_wgrad_wino(g, x, Cout, Cin, splits=splits)
    ref = _wgrad_ref(g, x)
    assert rel_l2(dw, ref) < TOL, rel_l2(dw, ref)
    assert float((dw.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    assert torch.equal(dw, ops.conv2d_wgrad_wino(g, x, Cout, Cin, splits=splits))
    with ops.conv3x3_algo("direct"):
        direct = ops.conv2d_wgrad(g, x, Cout, Cin, 3, 1)
    assert rel_l2(dw, direct) < TOL


def test_wgrad_wino_border_pixels_one_by_one(ops):
    """A one-hot output gradient at each border position (and a few inner ones) picks single input pixels: the zero padding
    (out-of-range LDS-DMA pieces) and the patch addressing, element by element."""
    dev = torch.device("cuda:0")
    B, Cin, Cout, H, W = 1, 64, 64, 8, 32
    x = torch.randn(B, Cin, H, W, generator=torch.Generator().manual_seed(5)).to(dev)
    spots = [(0, 0), (0, 15), (0, 16), (0, 31), (7, 0), (7, 31), (1, 1), (2, 17), (3, 15), (4, 16), (5, 30), (6, 2)]
    for h, w_ in spots:
        g = torch.zeros(B, Cout, H, W, device=dev)
        g[0, :, h, w_] = 1.0
        dw = ops.conv2d_wgrad_wino(g, x, Cout, Cin)
        xp = F.pad(x, (1, 1, 1, 1))[0, :, h:h + 3, w_:w_ + 3]                  # [Cin,3,3]
        assert float((dw - xp.unsqueeze(0)).abs().max()) < 1e-5, (h, w_)


def test_wgrad_wino_scale_accumulate_and_routing(ops):
    dev = torch.device("cuda:0")
    B, Cin, Cout, H, W = 4, 128, 64, 32, 32
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dev)
    g = torch.randn(B, Cout, H, W, generator=gen).to(dev)
    ref = _wgrad_ref(g, x)
    out = torch.ones(Cout, Cin, 3, 3, device=dev)
    ops.conv2d_wgrad_wino(g, x, Cout, Cin, scale=0.5, out=out, accumulate=True)
    assert rel_l2(out, 1.0 + 0.5 * ref) < TOL
    # ops.conv2d_wgrad routes a served plain 3x3 problem to the Winograd kernel (bitwise the same call) unless the switch says direct
    if ops.use_wgrad_wino(B, Cin, Cout, H, W):
        assert torch.equal(ops.conv2d_wgrad(g, x, Cout, Cin, 3, 1), ops.conv2d_wgrad_wino(g, x, Cout, Cin))
    # a x2 layer: the same gradient as the in-kernel-interpolating direct form
    xs = torch.randn(B, Cin, H // 2, W // 2, generator=gen).to(dev)
    a = ops.conv2d_wgrad(g, xs, Cout, Cin, 3, 1, upsample=True)
    with ops.conv3x3_algo("direct"):
        b = ops.conv2d_wgrad(g, xs, Cout, Cin, 3, 1, upsample=True)
    assert rel_l2(a, b) < TOL


def test_wgrad_wino_rejects_what_it_does_not_serve(ops):
    L = importlib.import_module("speak-hack_amd")._lib
    assert not ops.wgrad_wino_supported(2, 3, 64, 32, 32) and not ops.wgrad_wino_supported(2, 64, 96, 32, 32)
    assert not ops.wgrad_wino_supported(2, 64, 64, 32, 24) and not ops.wgrad_wino_supported(2, 64, 64, 7, 32)
    dev = torch.device("cuda:0")
    g, x = torch.zeros(2, 96, 32, 32, device=dev), torch.zeros(2, 64, 32, 32, device=dev)
    with pytest.raises(L.SpkError):
        ops.conv2d_wgrad_wino(g, x, 96, 64)


def test_wgrad_wino_modulated_equals_fp64(ops):
    """The modulated convolution's weight gradient (StyleGAN2 variant): x * s[b,ci] and g * d'[b,co] formed in registers on the
    way into the transforms -- against fp64 on explicitly rescaled operands, and, for a x2 layer (upfirdn2d [1,3,3,1] image),
    against the direct kernel that interpolates in LDS."""
    dev = torch.device("cuda:0")
    B, Cin, Cout, H, W = 3, 64, 128, 16, 32
    gen = torch.Generator().manual_seed(21)
    x = torch.randn(B, Cin, H, W, generator=gen).to(dev)
    g = torch.randn(B, Cout, H, W, generator=gen).to(dev)
    s = (torch.rand(B, Cin, generator=gen) + 0.5).to(dev)
    dp = (torch.rand(B, Cout, generator=gen) + 0.5).to(dev)
    dw = ops.conv2d_wgrad_wino(g, x, Cout, Cin, scale=0.7, batch_scale=s, g_scale=dp)
    ref = 0.7 * _wgrad_ref(g.double() * dp.double().view(B, Cout, 1, 1), x.double() * s.double().view(B, Cin, 1, 1))
    assert rel_l2(dw, ref) < TOL, rel_l2(dw, ref)
    xs = torch.randn(B, Cin, H // 2, W // 2, generator=gen).to(dev)
    a = ops.conv2d_wgrad_wino(g, ops.upsample2x(xs, zero_border=True), Cout, Cin, batch_scale=s, g_scale=dp)
    with ops.conv3x3_algo("direct"):
        b = ops.conv2d_wgrad(g, xs, Cout, Cin, 3, 1, upsample=True, up_fir=True, batch_scale=s, g_scale=dp)
    assert rel_l2(a, b) < TOL, rel_l2(a, b)
    L = importlib.import_module("speak-hack_amd")._lib
    with pytest.raises(L.SpkError):
        ops.conv2d_wgrad_wino(g, x, Cout, Cin, batch_scale=s)


@pytest.mark.parametrize("B,G,fold,Cin,Cout,H,W,shared", [(2, 6, 2, 64, 64, 16, 32, False), (1, 3, 1, 128, 64, 8, 16, False),
                                                          (2, 2, 2, 64, 128, 6, 48, True), (3, 1, 1, 64, 64, 8, 32, False)])
def test_wgrad_wino_grouped_folded_with_a_folded_batchnorm_input(ops, B, G, fold, Cin, Cout, H, W, shared):
    """The trunk's form (model.py:60-62 + :84-90: G encoder passes as groups, a pass pair sharing weights folded, the conv's input =
    relu(bn(x)) formed from the saved pre-BatchNorm tensor): against fp64 on the explicit tensors -- in particular the ZERO padding
    of the activated input (an affine + ReLU of a padded zero is not zero)."""
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(G * 10 + fold + H)
    Cx = Cin if shared else G * Cin
    x = torch.randn(B, Cx, H, W, generator=gen).to(dev)
    g = torch.randn(B, G * Cout, H, W, generator=gen).to(dev)
    sc = (torch.rand(Cx, generator=gen) + 0.5).to(dev)
    sh = (torch.randn(Cx, generator=gen) * 0.5 + 0.3).to(dev)         # mostly positive: a padded zero would become relu(shift) > 0
    dw = ops.conv2d_wgrad_wino(g, x, Cout, Cin, in_affine=(sc, sh), groups=G, shared_input=shared, fold=fold, scale=0.5)
    xa = torch.relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    per = [_wgrad_ref(g[:, q * Cout:(q + 1) * Cout], xa[:, (0 if shared else q * Cin):(0 if shared else q * Cin) + Cin]) for q in range(G)]
    n = G // fold
    ref = 0.5 * torch.cat([sum(per[q + f * n] for f in range(fold)) for q in range(n)], 0)
    assert dw.shape == ref.shape and rel_l2(dw, ref) < TOL, rel_l2(dw, ref)
    with ops.conv3x3_algo("direct"):
        direct = ops.conv2d_wgrad(g, x, Cout, Cin, 3, 1, in_affine=(sc, sh), groups=G, shared_input=shared, fold=fold, scale=0.5)
    assert rel_l2(dw, direct) < TOL


# ---- 16 x 16 regions and the sliced contraction (few regions: partial sums through the split-K workspace) ------------------------
@pytest.mark.parametrize("B,Cin,Cout,H,W,ksplit", [(8, 512, 512, 16, 16, 0), (1, 64, 64, 16, 16, 0), (2, 64, 128, 48, 16, 1), (1, 128, 64, 32, 48, 2),
                                                   (1, 512, 512, 32, 32, 0), (2, 256, 64, 8, 64, 4), (1, 64, 200, 16, 32, 2)])
def test_wino_square_regions_and_sliced_contraction(ops, B, Cin, Cout, H, W, ksplit):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + Cin + Cout + H + W + ksplit)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
    bias, nw = torch.randn(Cout, generator=g).to(dev), torch.randn(Cout, generator=g).to(dev)
    noise = torch.randn(B, 1, H, W, generator=g).to(dev)
    style = torch.randn(B, 2 * Cout, generator=g).to(dev)
    base = torch.randn(B, Cout, H, W, generator=g).to(dev)
    assert ops.wino_supported(B, Cin, Cout, H, W)
    ks = ops.wino_ksplit(B, Cin, Cout, H, W, ksplit)
    assert ks >= 1 and (Cin // 8) % (2 * ks) == 0
    if ksplit == 0 and (B, H) in ((8, 16), (1, 32)):
        assert ks > 1                                   # the cases this form exists for: 64 / 32 (region, channel tile) pairs
    wp = ops.pack_conv_weight_wino(w)
    y = ops.conv3x3_wino(x, wp, Cout, ksplit=ksplit)
    ref, _ = _ref(x, w)
    assert rel_l2(y, ref) < TOL, rel_l2(y, ref)
    assert float((y.double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    # the whole epilogue, the pre-style output and accumulate -- in the kernel (ks = 1) or in the split-K finisher (ks > 1)
    pre = torch.empty(B, Cout, H, W, device=dev)
    y2 = ops.conv3x3_wino(x, wp, Cout, bias=bias, noise_w=nw, noise=noise, style=style, lrelu_slope=0.2, out_scale=0.7, out_pre=pre,
                          out=base.clone(), accumulate=True, ksplit=ksplit)
    ref2, ref_pre = _ref(x, w, bias=bias, noise_w=nw, noise=noise, style=style, slope=0.2, out_scale=0.7)
    assert rel_l2(y2, ref2 + base.double()) < TOL and rel_l2(pre, ref_pre) < TOL
    assert torch.equal(y, ops.conv3x3_wino(x, wp, Cout, ksplit=ksplit))


def test_wino_modulated_sliced(ops):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    B, Cin, Cout, H, W = 2, 128, 64, 16, 16
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
    s_ = (torch.rand(B, Cin, generator=g) + 0.5).to(dev)
    dm = (torch.rand(B, Cout, generator=g) + 0.5).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    for ks in (1, 2, 4):
        y = ops.conv3x3_wino(x, ops.pack_conv_weight_wino(w), Cout, bias=bias, lrelu_slope=0.2, act_gain=2 ** 0.5, out_scale=0.3,
                             batch_scale=s_, demod=dm, ksplit=ks)
        ref = F.conv2d(x.double() * s_.double().view(B, Cin, 1, 1), w.double(), padding=1) * 0.3 * dm.double().view(B, Cout, 1, 1)
        ref = F.leaky_relu(ref + bias.double().view(1, -1, 1, 1), 0.2) * 2 ** 0.5
        assert rel_l2(y, ref) < TOL, (ks, rel_l2(y, ref))


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 64, 64, 16, 64), (1, 128, 64, 32, 32), (2, 64, 48, 16, 16), (8, 64, 64, 64, 64)])
def test_wino_fused_torgb_equals_conv_then_1x1(ops, B, Cin, Cout, H, W):
    """SPK_EPI_TORGB: the 1x1 conv of styleganv1.py:607 inside the last launch's epilogue -- against fp64 conv -> epilogue -> 1x1, with
    and without the activation store."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev)
    bias, nw = torch.randn(Cout, generator=g).to(dev), torch.randn(Cout, generator=g).to(dev)
    noise = torch.randn(B, 1, H, W, generator=g).to(dev)
    style = torch.randn(B, 2 * Cout, generator=g).to(dev)
    rw = (torch.randn(3, Cout, 1, 1, generator=g) / Cout ** 0.5).to(dev)
    rb = torch.randn(3, generator=g).to(dev)
    wp = ops.pack_conv_weight_wino(w)
    ref, _ = _ref(x, w, bias=bias, noise_w=nw, noise=noise, style=style, slope=0.2)
    ref_rgb = F.conv2d(ref, rw.double(), rb.double())
    y, rgb = ops.conv3x3_wino(x, wp, Cout, bias=bias, noise_w=nw, noise=noise, style=style, lrelu_slope=0.2, rgb=(rw, rb))
    assert rel_l2(y, ref) < TOL and rel_l2(rgb, ref_rgb) < TOL, (rel_l2(y, ref), rel_l2(rgb, ref_rgb))
    none, rgb2 = ops.conv3x3_wino(x, wp, Cout, bias=bias, noise_w=nw, noise=noise, style=style, lrelu_slope=0.2, rgb=(rw, rb), store_out=False)
    assert none is None and torch.equal(rgb, rgb2)
    _, rgb3 = ops.conv3x3_wino(x, wp, Cout, rgb=(rw, None))                           # no epilogue stage, no bias
    assert rel_l2(rgb3, F.conv2d(_ref(x, w)[0], rw.double())) < TOL
    L = importlib.import_module("speak-hack_amd")._lib
    with pytest.raises(L.SpkError):
        ops.conv3x3_wino(x, wp, Cout, rgb=(rw, rb), accumulate=True, out=torch.zeros(B, Cout, H, W, device=dev))


def test_wino_list_pack_equals_single_packs_bit_for_bit(ops):
    """ops.prepack_wino: every stale image of a decoder pass in ONE launch (spk_conv2d_pack_weights_wino_list), both orientations --
    the bits of the one-at-a-time packer; a second call packs nothing; an in-place update re-packs."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    ws = [torch.randn(co, ci, 3, 3, generator=g).to(dev) for co, ci in ((64, 64), (64, 128), (200, 48), (512, 256))]
    pks = [ops.PackedConvWeight() for _ in ws]
    items = [(pk, w, tf) for pk, w in zip(pks, ws) for tf in (False, True)]
    ops.prepack_wino(items)
    for pk, w in zip(pks, ws):
        for tf in (False, True):
            got = pk._cache[("wino", tf)][1]
            assert pk.get_wino(w, transpose_flip=tf) is got                      # served from the cache the list call filled
            assert torch.equal(got, ops.pack_conv_weight_wino(w, transpose_flip=tf))
    before = [pk._cache[("wino", False)][1] for pk in pks]
    ops.prepack_wino(items)
    assert all(pk._cache[("wino", False)][1] is b for pk, b in zip(pks, before))
    ws[1].mul_(2.0)
    ops.prepack_wino(items)
    assert pks[1]._cache[("wino", True)][1] is not before[1] and torch.equal(pks[1].get_wino(ws[1]), ops.pack_conv_weight_wino(ws[1]))


@pytest.mark.parametrize("B,G,Cin,Cout,H,W,acc", [(2, 6, 64, 64, 16, 32, False), (1, 3, 128, 64, 16, 16, True), (2, 2, 64, 200, 8, 32, False)])
def test_wino_grouped_launch_equals_per_group_convs(ops, B, G, Cin, Cout, H, W, acc):
    """The encoders' 3x3 data gradients (model.py:60-62, :84-90: the G passes as groups of one launch): every group against its own fp64
    convolution, with accumulate, whatever slicing the launch picks."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + G + Cin + Cout)
    x = torch.randn(B, G * Cin, H, W, generator=g).to(dev)
    ws = [(torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dev) for _ in range(G)]
    n = ops.L.lib().spk_conv2d_packed_bytes_wino(Cin, Cout) // 4
    wp = torch.empty(G * n, device=dev)
    ops.pack_conv_weights_wino_into(ws, [wp[i * n:(i + 1) * n] for i in range(G)])
    base = torch.randn(B, G * Cout, H, W, generator=g).to(dev)
    y = ops.conv3x3_wino(x, wp, Cout, groups=G, out=base.clone() if acc else None, accumulate=acc)
    ref = torch.cat([F.conv2d(x[:, q * Cin:(q + 1) * Cin].double(), ws[q].double(), padding=1) for q in range(G)], 1)
    if acc:
        ref = ref + base.double()
    assert rel_l2(y, ref) < TOL, rel_l2(y, ref)
    L = importlib.import_module("speak-hack_amd")._lib
    with pytest.raises(L.SpkError):
        ops.conv3x3_wino(x, wp, Cout, groups=G, lrelu_slope=0.2)
