"""Grouped launches of the conv kernel (several independent convs of one shape per launch -- how the three IRFD
encoders, which run the same ResNet-50 on the same image, share their launches): bitwise equal to the separate launches
they replace, for every kernel family, the shared-input stem form, the folded BatchNorm affine, the BatchNorm sums,
split-K shapes and the data-gradient (transpose-flip) packing."""
import importlib

import pytest
import torch

from oracle.weights_recipe import recipe_input, recipe_tensor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    return importlib.import_module("speak-hack_amd.ops")


@pytest.mark.parametrize("k,stride,B,Cin,Cout,H,shared,aff", [
    (1, 1, 2, 64, 256, 16, False, True),     # bottleneck 1x1 (folded BN + ReLU on the way in)
    (3, 1, 2, 64, 64, 16, False, True),      # bottleneck 3x3
    (3, 2, 2, 128, 128, 16, False, True),    # strided 3x3
    (1, 2, 2, 256, 512, 16, False, False),   # downsample 1x1 stride 2
    (7, 2, 2, 3, 64, 32, True, False),       # stem: every group reads the same image
    (1, 1, 8, 1024, 2048, 4, False, True),   # layer4 shape: few pixels, split-K
    (3, 1, 1, 20, 24, 9, False, False),      # ragged channels, odd size
])
def test_grouped_equals_separate(ops, k, stride, B, Cin, Cout, H, shared, aff):
    dev, G = torch.device("cuda:0"), 3
    torch.manual_seed(0)
    xs = [torch.randn(B, Cin, H, H, device=dev) for _ in range(1 if shared else G)]
    ws = [torch.randn(Cout, Cin, k, k, device=dev) * 0.05 for _ in range(G)]
    sc = [torch.rand(Cin, device=dev) + 0.5 for _ in range(G)] if aff else None
    sh = [torch.randn(Cin, device=dev) * 0.1 for _ in range(G)] if aff else None
    Ho = ops.conv_out_size(H, k, stride)
    cfg = ops.conv2d_pick_config(k, stride, B, Cin, Cout, Ho, Ho)
    # the automatic split-K factor looks at the whole grid (3x the tiles when grouped): pin it, so that the summation
    # order -- and with it every bit -- is the same in both forms
    ks = 4 if Cin >= 512 else 1
    sep, sep_stats = [], []
    for g in range(G):
        st = torch.zeros(2 * Cout, device=dev, dtype=torch.float64)
        y = ops.conv2d_fused(xs[0 if shared else g], ops.pack_conv_weight(ws[g], cfg), Cout, k, stride, stats=st, config=cfg,
                             in_affine=(sc[g], sh[g]) if aff else None, ksplit=ks)
        sep.append(y)
        sep_stats.append(st)
    x_all = xs[0] if shared else torch.cat(xs, dim=1).contiguous()
    wp_all = torch.cat([ops.pack_conv_weight(ws[g], cfg) for g in range(G)])
    st_all = torch.zeros(2 * G * Cout, device=dev, dtype=torch.float64)
    y_all = ops.conv2d_fused(x_all, wp_all, Cout, k, stride, stats=st_all, config=cfg, groups=G, shared_input=shared,
                             in_affine=(torch.cat(sc), torch.cat(sh)) if aff else None, ksplit=ks)
    assert y_all.shape == (B, G * Cout, Ho, Ho)
    assert torch.equal(y_all, torch.cat(sep, dim=1))
    sums = torch.cat([s[:Cout] for s in sep_stats] + [s[Cout:] for s in sep_stats])
    assert torch.allclose(st_all, sums, rtol=1e-12, atol=0)       # fp64 atomics: order of addition may differ


def test_grouped_data_gradient(ops):
    """dgrad of a grouped conv = the grouped conv of the transpose-flipped weights (groups keep their channels)."""
    dev, G, B, Cin, Cout, H = torch.device("cuda:0"), 3, 2, 64, 128, 16
    torch.manual_seed(1)
    g_out = torch.randn(B, G * Cout, H, H, device=dev)
    ws = [torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05 for _ in range(G)]
    cfg = ops.conv2d_pick_config(3, 1, B, Cout, Cin, H, H)
    sep = [ops.conv2d_fused(g_out[:, g * Cout:(g + 1) * Cout].contiguous(), ops.pack_conv_weight(ws[g], cfg, transpose_flip=True),
                            Cin, 3, 1, config=cfg) for g in range(G)]
    wp = torch.cat([ops.pack_conv_weight(ws[g], cfg, transpose_flip=True) for g in range(G)])
    dx = ops.conv2d_fused(g_out, wp, Cin, 3, 1, config=cfg, groups=G)
    assert torch.equal(dx, torch.cat(sep, dim=1))


@pytest.mark.parametrize("k,stride,B,Cin,Cout,H,shared,aff", [
    (1, 1, 2, 64, 256, 16, False, True),
    (3, 1, 2, 64, 64, 16, False, True),
    (3, 2, 2, 128, 128, 16, False, False),
    (7, 2, 2, 3, 64, 32, True, False),
    (1, 1, 4, 1024, 2048, 4, False, True),
])
def test_grouped_weight_gradient(ops, k, stride, B, Cin, Cout, H, shared, aff):
    """Grouped wgrad = the groups' weight gradients one after another (bitwise, with the pixel split pinned)."""
    dev, G = torch.device("cuda:0"), 3
    torch.manual_seed(2)
    Ho = ops.conv_out_size(H, k, stride)
    xs = [torch.randn(B, Cin, H, H, device=dev) for _ in range(1 if shared else G)]
    gs = [torch.randn(B, Cout, Ho, Ho, device=dev) for _ in range(G)]
    sc = [torch.rand(Cin, device=dev) + 0.5 for _ in range(G)] if aff else None
    sh = [torch.randn(Cin, device=dev) * 0.1 for _ in range(G)] if aff else None
    sep = [ops.conv2d_wgrad(gs[q], xs[0 if shared else q], Cout, Cin, k, stride, splits=2,
                            in_affine=(sc[q], sh[q]) if aff else None) for q in range(G)]
    x_all = xs[0] if shared else torch.cat(xs, dim=1).contiguous()
    dw = ops.conv2d_wgrad(torch.cat(gs, dim=1).contiguous(), x_all, Cout, Cin, k, stride, splits=2, groups=G, shared_input=shared,
                          in_affine=(torch.cat(sc), torch.cat(sh)) if aff else None)
    assert dw.shape == (G * Cout, Cin, k, k)
    assert torch.equal(dw, torch.cat(sep, dim=0))


@pytest.mark.parametrize("k,stride,B,Cin,Cout,H", [(1, 1, 2, 64, 256, 16), (3, 1, 2, 64, 64, 16), (3, 2, 2, 128, 128, 16),
                                                     (7, 2, 2, 3, 64, 32), (1, 1, 4, 1024, 2048, 4), (3, 1, 2, 64, 128, 24)])
def test_grouped_weight_gradient_folded_over_images(ops, k, stride, B, Cin, Cout, H):
    """``fold`` = 2: groups q and q + 3 are the same conv on a second image (IRFD runs every encoder on x_s and x_t);
    the slab reduce adds their gradients: equal to the sum of the two halves of the unfolded result."""
    dev, G = torch.device("cuda:0"), 6
    torch.manual_seed(3)
    Ho = ops.conv_out_size(H, k, stride)
    x = torch.randn(B, G * Cin, H, H, device=dev)
    g = torch.randn(B, G * Cout, Ho, Ho, device=dev)
    full = ops.conv2d_wgrad(g, x, Cout, Cin, k, stride, splits=2, groups=G)
    folded = ops.conv2d_wgrad(g, x, Cout, Cin, k, stride, splits=2, groups=G, fold=2)
    assert folded.shape == (3 * Cout, Cin, k, k)
    want = full.view(2, 3 * Cout, Cin, k, k).sum(0)
    assert float((folded - want).abs().max()) <= 1e-5 * float(want.abs().max())
    with pytest.raises(Exception):
        ops.conv2d_wgrad(g, x, Cout, Cin, k, stride, groups=G, fold=4)


@pytest.mark.gpu
@pytest.mark.parametrize("k,Cin,Cout,cfg", [(1, 64, 256, 12), (1, 256, 64, 10), (3, 64, 64, 6), (3, 36, 40, 2), (7, 3, 64, 4), (3, 64, 48, 13)])
@pytest.mark.parametrize("n", [1, 3, 6, 11])
def test_pack_list_equals_separate_packs(ops, k, Cin, Cout, cfg, n):
    """``spk_conv2d_pack_weights_list`` (several weights of a grouped launch on one launch, also past the 8 of one call)
    writes exactly the images of the per-tensor entry point, forward and transposed; an ``out`` buffer is refilled in place."""
    dev = torch.device("cuda:0")
    ws = [torch.randn(Cout, Cin, k, k, device=dev) for _ in range(n)]
    for tf in ([2] if cfg == 13 else [0, 1, 2] if (k == 3 and cfg < 4) else [0, 1] if k != 7 else [0]):
        ref = torch.cat([ops.pack_conv_weight(w, cfg, tf) for w in ws])
        got = ops.pack_conv_weights_list(ws, cfg, tf)
        assert torch.equal(got, ref)
        again = ops.pack_conv_weights_list([w * 2 for w in ws], cfg, tf, out=got)
        assert again.data_ptr() == got.data_ptr() and torch.equal(again, ref * 2)
