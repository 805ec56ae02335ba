"""GPU parity of the ResNet-50 trunk path: generic conv (1x1 / 3x3 / 7x7,
stride 1/2, folded BatchNorm+ReLU input, BatchNorm statistics in the epilogue), BN finalize, the
residual pass, pools, and the whole trunk in eval and train mode against the CPU oracle.
Tolerances: 2e-5 rel-L2 per op, 2e-4 through the 53-conv trunk (exact fp32 arithmetic; only the
summation order and the folded-BN rounding differ)."""
import importlib

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import resnet_ref as RR
from oracle.weights_recipe import recipe_input, recipe_tensor, resnet_trunk_state_dict

pytestmark = pytest.mark.gpu
TOL_OP = 2e-5
TOL_TRUNK = 2e-4


@pytest.fixture(scope="module")
def pkg():
    assert torch.cuda.is_available()
    p = importlib.import_module("speak-hack_amd")
    p._lib.lib()
    return p


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("k,stride,B,Cin,Cout,H,W", [
    (1, 1, 2, 64, 256, 16, 16), (1, 1, 3, 40, 72, 9, 13), (1, 1, 1, 256, 64, 64, 64), (1, 1, 2, 2048, 512, 8, 8),
    (1, 2, 2, 256, 512, 16, 16), (1, 2, 1, 24, 40, 15, 11),
    (3, 2, 2, 128, 128, 32, 32), (3, 2, 3, 20, 36, 13, 10), (3, 2, 1, 512, 512, 16, 16),
    (7, 2, 2, 3, 64, 64, 64), (7, 2, 1, 3, 64, 37, 51),
])
def test_conv2d_kernels_strides(pkg, dev, k, stride, B, Cin, Cout, H, W):
    tag = f"c2.{k}.{stride}.{B}.{Cin}.{Cout}.{H}.{W}"
    x = recipe_input(tag + ".x", (B, Cin, H, W))
    w = recipe_tensor(tag + ".weight", (Cout, Cin, k, k))
    ref = F.conv2d(x, w, stride=stride, padding=(k - 1) // 2)
    Ho, Wo = ref.shape[-2:]
    n = pkg._lib.lib().spk_conv2d_num_configs()
    ran = 0
    for cfg in [-1] + list(range(n)):
        c = cfg if cfg >= 0 else pkg.ops.conv2d_pick_config(k, stride, B, Cin, Cout, Ho, Wo)
        if not pkg.ops.conv2d_config_fits(c, k, stride, B, Cin, Cout, Ho, Wo):
            continue
        wp = pkg.ops.pack_conv_weight(w.to(dev), c)
        for ksplit in (1, 0):
            y = pkg.ops.conv2d_fused(x.to(dev), wp, Cout, k, stride, config=c, ksplit=ksplit)
            assert rel_l2(y, ref) < TOL_OP, (c, ksplit)
        ran += 1
    assert ran >= 2


@pytest.mark.parametrize("B,Hin,Win,G,shared", [(2, 64, 64, 1, False), (3, 50, 72, 2, False), (2, 34, 136, 3, True), (1, 256, 256, 2, True)])
def test_stem_form_7x7_stride2(pkg, dev, B, Hin, Win, G, shared):
    """Config 16 (conv7x7_stem.hip: K = 147 implicit GEMM on v_mfma_f32_32x32x1_2b, parity-split input patch in LDS): partial
    tiles in both directions, grouped with own / shared images, BatchNorm sums from the epilogue (own slots and shared slot)."""
    tag = f"stem.{B}.{Hin}.{Win}.{G}.{int(shared)}"
    x = recipe_input(tag + ".x", (B, 3 if shared else 3 * G, Hin, Win), "uniform")
    ws = [recipe_tensor(tag + f".weight{q}", (64, 3, 7, 7)) for q in range(G)]
    ref = torch.cat([F.conv2d(x if shared else x[:, 3 * q:3 * q + 3], ws[q], stride=2, padding=3) for q in range(G)], 1)
    Ho, Wo = ref.shape[-2:]
    cfg = pkg.ops.conv2d_pick_config(7, 2, B, 3, 64, Ho, Wo)
    assert cfg == 16
    wp = pkg.ops.pack_conv_weights_list([w.to(dev) for w in ws], cfg)
    slots = pkg.ops.stats_slots(cfg, 7, 2, B, 3, 64, Ho, Wo)
    r64 = ref.double()
    sums = torch.cat([r64.sum((0, 2, 3)), (r64 * r64).sum((0, 2, 3))])
    for ns in (slots, 1):
        stats = torch.zeros(ns * 2 * G * 64, device=dev, dtype=torch.float64)
        y = pkg.ops.conv2d_fused(x.to(dev), wp, 64, 7, 2, stats=stats, config=cfg, groups=G, shared_input=shared)
        assert rel_l2(y, ref) < TOL_OP
        got = stats.view(ns, -1).sum(0).cpu()
        assert torch.allclose(got, sums, rtol=2e-5, atol=1e-3), (got - sums).abs().max()
    y = pkg.ops.conv2d_fused(x.to(dev), wp, 64, 7, 2, config=cfg, groups=G, shared_input=shared)      # no statistics
    assert rel_l2(y, ref) < TOL_OP


@pytest.mark.parametrize("k,stride", [(1, 1), (3, 1), (3, 2), (1, 2)])
def test_conv2d_folded_bn_relu_input_and_statistics(pkg, dev, k, stride):
    """x' = relu(x*a+b) applied while staging; sum / sum-of-squares of y from the epilogue."""
    B, Cin, Cout, H = 3, 44, 72, 14
    tag = f"c2aff.{k}.{stride}"
    x = recipe_input(tag + ".x", (B, Cin, H, H))
    a = 1.0 + recipe_tensor(tag + ".a", (Cin,), 0.3)
    b = recipe_tensor(tag + ".b", (Cin,), 0.3)
    w = recipe_tensor(tag + ".weight", (Cout, Cin, k, k))
    xin = F.relu(x * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1))
    ref = F.conv2d(xin, w, stride=stride, padding=(k - 1) // 2)
    cfg = pkg.ops.conv2d_pick_config(k, stride, B, Cin, Cout, ref.shape[-2], ref.shape[-1])
    wp = pkg.ops.pack_conv_weight(w.to(dev), cfg)
    for ksplit in (1, 2):
        stats = torch.zeros(2 * Cout, device=dev, dtype=torch.float64)
        y = pkg.ops.conv2d_fused(x.to(dev), wp, Cout, k, stride, in_affine=(a.to(dev), b.to(dev)), stats=stats,
                                 config=cfg, ksplit=ksplit)
        assert rel_l2(y, ref) < TOL_OP
        assert rel_l2(stats[:Cout], ref.double().sum((0, 2, 3))) < 1e-6
        assert rel_l2(stats[Cout:], (ref.double() ** 2).sum((0, 2, 3))) < 1e-6


@pytest.mark.parametrize("slots,groups", [(4, 1), (16, 1), (8, 3), (3, 1), (0, 1), (0, 3)])
def test_conv2d_statistics_spread_over_slots(pkg, dev, slots, groups):
    """``stats_slots`` copies of the sums (pixel tile i -> copy i % slots; 0 here = one copy per tile, the plain-store
    form): their total, and the affine ``bn_finalize`` makes of them, equal the single-copy result; ragged tiles,
    ordinary and grouped launches."""
    B, Cin, Cout, H, W = 5, 20, 40, 30, 22
    tag = f"c2slots.{slots}.{groups}"
    x = recipe_input(tag + ".x", (B, groups * Cin, H, W))
    ws = [recipe_tensor(tag + f".weight{g}", (Cout, Cin, 3, 3)) for g in range(groups)]
    ref = torch.cat([F.conv2d(x[:, g * Cin:(g + 1) * Cin], ws[g], padding=1) for g in range(groups)], 1)
    cfg = pkg.ops.conv2d_pick_config(3, 1, B, Cin, Cout, H, W)
    wp = torch.cat([pkg.ops.pack_conv_weight(w.to(dev), cfg) for w in ws])
    Cy = groups * Cout
    if slots == 0:              # one copy per pixel tile: plain stores, no atomics
        slots = pkg.ops.stats_slots(cfg, 3, 1, B, Cin, Cout, H, W)
        assert slots > 4
    stats = torch.zeros(slots * 2 * Cy, device=dev, dtype=torch.float64)
    y = pkg.ops.conv2d_fused(x.to(dev), wp, Cout, 3, 1, stats=stats, config=cfg, groups=groups)
    assert rel_l2(y, ref) < TOL_OP
    per = stats.view(slots, 2, Cy)
    assert int((per.abs().sum((1, 2)) > 0).sum()) > 1                     # more than one copy was written
    assert rel_l2(per.sum(0)[0], ref.double().sum((0, 2, 3))) < 1e-6
    assert rel_l2(per.sum(0)[1], (ref.double() ** 2).sum((0, 2, 3))) < 1e-6
    g, be = 1.0 + recipe_tensor(tag + ".g", (Cy,), 0.2), recipe_tensor(tag + ".b", (Cy,), 0.2)
    rm, rv = torch.zeros(Cy), torch.ones(Cy)
    want = F.batch_norm(ref, rm.clone(), rv.clone(), g, be, True, 0.1, 1e-5)
    rm_d, rv_d = rm.to(dev), rv.to(dev)
    sc, sh = pkg.ops.bn_finalize(stats, B * H * W, g.to(dev), be.to(dev), rm_d, rv_d, 0.1, 1e-5)
    assert rel_l2(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), want) < TOL_OP
    F.batch_norm(ref, rm, rv, g, be, True, 0.1, 1e-5)
    assert rel_l2(rm_d, rm) < 1e-5 and rel_l2(rv_d, rv) < 1e-5
    # the totals are in copy 0 now: the replayed update of a pass that ran twice reads one copy
    sc2, sh2 = pkg.ops.bn_finalize(stats[:2 * Cy], B * H * W, g.to(dev), be.to(dev), rm_d, rv_d, 0.1, 1e-5)
    assert torch.equal(sc2, sc) and torch.equal(sh2, sh)
    F.batch_norm(ref, rm, rv, g, be, True, 0.1, 1e-5)
    assert rel_l2(rm_d, rm) < 1e-5 and rel_l2(rv_d, rv) < 1e-5
    with pytest.raises(pkg._lib.SpkError):
        pkg.ops.conv2d_fused(x.to(dev), wp, Cout, 3, 1, stats=torch.zeros(3 * 2 * Cy + 1, device=dev, dtype=torch.float64),
                             config=cfg, groups=groups)


def test_bn_finalize_train_and_eval(pkg, dev):
    C, B, H = 37, 4, 9
    y = recipe_input("bnf.y", (B, C, H, H)) * 2.0 + 0.7
    g, be = 1.0 + recipe_tensor("bnf.g", (C,), 0.2), recipe_tensor("bnf.b", (C,), 0.2)
    rm, rv = recipe_tensor("bnf.rm", (C,), 0.3), recipe_tensor("bnf.rv", (C,), 1.0).abs() + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.batch_norm(y, rm_ref, rv_ref, g, be, True, 0.1, 1e-5)
    stats = torch.cat([y.double().sum((0, 2, 3)), (y.double() ** 2).sum((0, 2, 3))]).to(dev)
    rm_d, rv_d = rm.to(dev), rv.to(dev)
    sc, sh, mean, invstd = pkg.ops.bn_finalize(stats, B * H * H, g.to(dev), be.to(dev), rm_d, rv_d, 0.1, 1e-5, save=True)
    out = y.to(dev) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    assert rel_l2(out, ref) < TOL_OP
    assert rel_l2(rm_d, rm_ref) < 1e-6 and rel_l2(rv_d, rv_ref) < 1e-6
    assert rel_l2(mean, y.mean((0, 2, 3))) < 1e-6
    assert rel_l2(invstd, 1.0 / torch.sqrt(y.var((0, 2, 3), unbiased=False) + 1e-5)) < 1e-5
    # eval: running statistics, nothing updated
    ref_e = F.batch_norm(y, rm_ref, rv_ref, g, be, False, 0.1, 1e-5)
    sc, sh = pkg.ops.bn_finalize(None, 1, g.to(dev), be.to(dev), rm_d, rv_d, 0.0, 1e-5)
    assert rel_l2(y.to(dev) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), ref_e) < TOL_OP
    assert rel_l2(rm_d, rm_ref) < 1e-6


def test_residual_pass_and_pools(pkg, dev):
    B, C, H, W = 2, 19, 11, 14
    a, b = recipe_input("rp.a", (B, C, H, W)), recipe_input("rp.b", (B, C, H, W))
    sa, ba, sb, bb = (recipe_tensor(f"rp.{n}", (C,), 0.5) for n in ("sa", "ba", "sb", "bb"))
    v = lambda t: t.view(1, -1, 1, 1)
    ref = F.relu(a * v(sa) + v(ba) + b * v(sb) + v(bb))
    out = pkg.ops.bn_add_relu(a.to(dev), sa.to(dev), ba.to(dev), b.to(dev), sb.to(dev), bb.to(dev))
    assert rel_l2(out, ref) < 1e-6
    ref = F.relu(a * v(sa) + v(ba) + b)
    assert rel_l2(pkg.ops.bn_add_relu(a.to(dev), sa.to(dev), ba.to(dev), b.to(dev)), ref) < 1e-6
    x4 = recipe_input("rp.x4", (2, 8, 16, 20))       # HW % 4 == 0: vector path
    assert rel_l2(pkg.ops.bn_add_relu(x4.to(dev), None, None, x4.to(dev), relu=False), 2 * x4) < 1e-6
    # max pool 3x3 s2 p1, plain and with folded affine+relu; odd sizes
    for shape in [(2, 5, 9, 12), (1, 3, 16, 16), (1, 2, 1, 1)]:
        x = recipe_input(f"rp.mp.{shape}", shape)
        assert rel_l2(pkg.ops.maxpool3x3s2(x.to(dev)), F.max_pool2d(x, 3, 2, 1)) < 1e-7
        s, o = 1.0 + recipe_tensor("rp.mp.s", (shape[1],), 0.3), recipe_tensor("rp.mp.o", (shape[1],), 0.3)
        ref = F.max_pool2d(F.relu(x * v(s) + v(o)), 3, 2, 1)
        assert rel_l2(pkg.ops.maxpool3x3s2(x.to(dev), s.to(dev), o.to(dev)), ref) < 1e-6
    x = recipe_input("rp.ap", (3, 7, 5, 9))
    assert rel_l2(pkg.ops.global_avgpool(x.to(dev)), F.adaptive_avg_pool2d(x, 1)) < 1e-6


def _trunk(pkg, dev, prefix):
    enc = importlib.import_module("speak-hack_amd.encoder")
    m = enc.ResNet50Trunk()
    sd = resnet_trunk_state_dict(prefix)
    m.load_state_dict(sd)
    return m.to(dev), sd


def test_trunk_state_dict_keys(pkg, dev):
    enc = importlib.import_module("speak-hack_amd.encoder")
    m = enc.ResNet50Trunk()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == RR.trunk_param_shapes()
    assert sum(p.numel() for p in m.parameters()) == 23508032


@pytest.mark.parametrize("B,H", [(2, 96), (2, 256), (1, 64)])
def test_trunk_eval_vs_oracle(pkg, dev, B, H):
    m, sd = _trunk(pkg, dev, "Ei.")
    m.eval()
    x = recipe_input(f"trunk.x.{B}.{H}", (B, 3, H, H), "uniform")
    with torch.no_grad():
        y = m(x.to(dev))
        ref = RR.resnet50_trunk(x, sd)
    assert y.shape == (B, 2048, 1, 1)
    assert rel_l2(y, ref) < TOL_TRUNK


def test_trunk_train_mode_batch_statistics_and_running_update(pkg, dev):
    m, sd = _trunk(pkg, dev, "Ee.")
    m.train()
    x = recipe_input("trunk.train.x", (4, 3, 128, 128), "uniform")
    with torch.no_grad():
        y = m(x.to(dev))
        ref = RR.resnet50_trunk(x, sd, training=True, update_running_stats=True)
    assert rel_l2(y, ref) < TOL_TRUNK
    got = m.state_dict()
    for k in ("1.running_mean", "1.running_var", "4.0.bn2.running_var", "5.0.downsample.1.running_mean",
              "7.2.bn3.running_var", "7.2.bn3.running_mean"):
        assert rel_l2(got[k], sd[k]) < 1e-4, k
    assert int(got["6.3.bn1.num_batches_tracked"]) == 1


@pytest.mark.parametrize("cfg", [12, 14, 15])
@pytest.mark.parametrize("B,Cin,Cout,H,W,groups", [
    (2, 64, 256, 8, 8, 1),       # 64-pixel planes: a 128-pixel tile spans two images
    (3, 20, 72, 8, 12, 1),       # ragged channels (Cin % 32 != 0, Cout % 128 != 0), 288 pixels: a partial last tile
    (1, 256, 64, 16, 16, 1),     # one co tile half empty
    (2, 36, 136, 4, 8, 3),       # grouped, two co tiles per group (the second ragged)
    (8, 128, 512, 32, 32, 1),    # a trunk shape
    (3, 24, 40, 4, 4, 2),        # a ragged second k-tile (moved back to end at Cin), 16-pixel planes
    (2, 16, 8, 2, 2, 1),         # one k-tile, one group of 8 rows, 4-pixel planes
])
def test_conv1x1_gemm_form(pkg, dev, B, Cin, Cout, H, W, groups, cfg):
    """Tile configs 12 / 14 / 15: the stride-1 1x1 conv as a plain GEMM (128co- or 64co- x 128px blocks over the flattened
    pixel axis; packed weight = the [Cout][Cin] matrix for 12, [co tile][k tile][16][CO_T] for 14 / 15), with every
    epilogue / staging option it carries, against F.conv2d."""
    ops, L = pkg.ops, pkg._lib
    if cfg == 12 and (H * W) % 32:
        pytest.skip("config 12 needs H*W % 32 == 0")
    if cfg != 12:      # what the lean form asks for: >= 16 contraction channels in fours, output channels in eights
        assert not ops.conv2d_config_fits(cfg, 1, 1, B, 8, Cout, H, W) and not ops.conv2d_config_fits(cfg, 1, 1, B, Cin, Cout + 4, H, W)
    assert ops.conv2d_config_fits(cfg, 1, 1, B, Cin, Cout, H, W) and not ops.conv2d_config_fits(cfg, 3, 1, B, Cin, Cout, H, W)
    tag = f"g1x1.{B}.{Cin}.{Cout}.{H}.{W}.{groups}"
    G = groups
    x = recipe_input(tag + ".x", (B, G * Cin, H, W))
    ws = [recipe_tensor(tag + f".w{q}", (Cout, Cin, 1, 1)) for q in range(G)]
    bias = recipe_tensor(tag + ".bias", (G * Cout,), 0.3)
    a = 1.0 + recipe_tensor(tag + ".a", (G * Cin,), 0.3)
    b = recipe_tensor(tag + ".b", (G * Cin,), 0.3)
    base = recipe_input(tag + ".base", (B, G * Cout, H, W))

    def ref(xin, with_bias, slope):
        y = torch.cat([F.conv2d(xin[:, q * Cin:(q + 1) * Cin], ws[q]) for q in range(G)], 1)
        if with_bias:
            y = y + bias.view(1, -1, 1, 1)
        return F.leaky_relu(y, slope) if slope is not None else y

    wp = torch.cat([ops.pack_conv_weight(w.to(dev), cfg) for w in ws])
    if cfg == 12:
        assert wp.numel() == G * Cout * Cin
    else:
        co_t = 128 if cfg == 14 else 64
        assert wp.numel() == G * (-(-Cout // co_t)) * (-(-Cin // 16)) * 16 * co_t
        assert torch.equal(wp, ops.pack_conv_weights_list([w.to(dev) for w in ws], cfg))
    xd = x.to(dev)
    # plain
    assert rel_l2(ops.conv2d_fused(xd, wp, Cout, 1, 1, config=cfg, groups=G), ref(x, False, None)) < TOL_OP
    # bias + LeakyReLU
    y = ops.conv2d_fused(xd, wp, Cout, 1, 1, config=cfg, groups=G, bias=bias.to(dev), lrelu_slope=0.2)
    assert rel_l2(y, ref(x, True, 0.2)) < TOL_OP
    # folded BatchNorm + ReLU on the way in, BatchNorm sums on the way out (one copy per pixel tile), accumulate
    xin = F.relu(x * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1))
    want = ref(xin, False, None)
    slots = ops.stats_slots(cfg, 1, 1, B, Cin, Cout, H, W)
    assert slots == min(-(-B * H * W // 128), 2048)
    for s in (slots, 1):
        stats = torch.zeros(s * 2 * G * Cout, device=dev, dtype=torch.float64)
        out = base.to(dev).clone()
        y = ops.conv2d_fused(xd, wp, Cout, 1, 1, config=cfg, groups=G, in_affine=(a.to(dev), b.to(dev)), stats=stats, out=out,
                             accumulate=True)
        assert y.data_ptr() == out.data_ptr() and rel_l2(y, want + base) < TOL_OP
        tot = stats.view(s, 2, G * Cout).sum(0)
        full = (want + base).double()
        assert rel_l2(tot[0], full.sum((0, 2, 3))) < 1e-6 and rel_l2(tot[1], (full ** 2).sum((0, 2, 3))) < 1e-6
    # the transposed pack = the data gradient
    if G == 1:
        xg = x.clone().requires_grad_(True)
        g = recipe_input(tag + ".g", (B, Cout, H, W))
        F.conv2d(xg, ws[0]).backward(g)
        if ops.conv2d_config_fits(cfg, 1, 1, B, Cout, Cin, H, W):
            wt = ops.pack_conv_weight(ws[0].to(dev), cfg, transpose_flip=True)
            assert rel_l2(ops.conv2d_fused(g.to(dev), wt, Cin, 1, 1, config=cfg), xg.grad) < TOL_OP
    # the half-resolution accumulate: y(2h, 2w) += t(h, w) -- a stride-2 1x1 conv's data gradient joining without dilation
    if cfg != 12 and W % 4 == 0:
        t = recipe_input(tag + ".half", (B, G * Cout, (H + 1) // 2, (W + 1) // 2))
        dil = torch.zeros(B, G * Cout, H, W)
        dil[:, :, ::2, ::2] = t
        y = ops.conv2d_fused(xd, wp, Cout, 1, 1, config=cfg, groups=G, bias=bias.to(dev), accum_half=t.to(dev))
        assert rel_l2(y, ref(x, True, None) + dil) < TOL_OP
    elif cfg == 12:
        with pytest.raises(L.SpkError):
            ops.conv2d_fused(xd, wp, Cout, 1, 1, config=cfg, groups=G, accum_half=torch.zeros(B, G * Cout, (H + 1) // 2, (W + 1) // 2, device=dev))
    # what it does not carry is refused, not mis-computed
    with pytest.raises(L.SpkError):
        ops.conv2d_fused(xd, wp, Cout, 1, 1, config=cfg, groups=G, noise_w=bias.to(dev), noise=torch.zeros(B, 1, H, W, device=dev))
