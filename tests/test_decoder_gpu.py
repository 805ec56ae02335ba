"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs, and against the committed golden vectors produced by the reference.

Tolerance: the conv runs on the exact-fp32 MFMA pipe (an fmaf chain), so the only difference from
the oracle is summation order: TOL = 2e-5 rel-L2 per op / block, 1e-4 end to end (13 stacked
layers) -- the north-star bound is 1e-3.
"""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import decoder_ref as R
from oracle.weights_recipe import fill_state_dict, recipe_input, recipe_noises, recipe_tensor

pytestmark = pytest.mark.gpu
TOL_OP = 2e-5
TOL_E2E = 1e-4


@pytest.fixture(scope="module")
def pkg():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    p = importlib.import_module("speak-hack_amd")
    p._lib.lib()  # must load: no fallback
    return p


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Cin,Cout,H,W", [
    (2, 16, 8, 12, 12),      # ragged: H,W not powers of two, Cout < tile, Cin = 2 chunks
    (1, 3, 5, 7, 9),         # tiny odd everything, Cin < CI_T
    (3, 24, 40, 8, 8),       # several images per pixel tile, batch not a multiple of TB
    (2, 64, 64, 32, 32),
    (1, 128, 64, 64, 64),
    (2, 20, 130, 16, 40),    # Cout just over one 128-tile, non-square
    (1, 8, 32, 4, 4),        # 4x4 (ProGAN initial conv)
])
@pytest.mark.parametrize("config", [-1, 0, 1, 2, 3, 4, 5])
def test_conv3x3_plain_all_configs(pkg, dev, B, Cin, Cout, H, W, config):
    x = recipe_input(f"cv.x.{B}.{Cin}.{H}.{W}", (B, Cin, H, W))
    w = recipe_tensor(f"cv.{Cout}.{Cin}.weight", (Cout, Cin, 3, 3))
    b = recipe_tensor(f"cv.{Cout}.bias", (Cout,))
    ref = F.conv2d(x, w, b, padding=1)
    cfg = config if config >= 0 else pkg.ops.conv3x3_pick_config(B, Cin, Cout, H, W)
    if not pkg.ops.conv3x3_config_fits(cfg, B, Cin, Cout, H, W):
        pytest.skip("tile config cannot host this shape (the auto pick never selects it)")
    wp = pkg.ops.pack_conv3x3_weight(w.to(dev), cfg)
    for ksplit in (1, 0, 3):     # none / auto / forced odd split (ragged last slice)
        y = pkg.ops.conv3x3_fused(x.to(dev), wp, Cout, bias=b.to(dev), config=cfg, ksplit=ksplit)
        assert rel_l2(y, ref) < TOL_OP, ksplit


@pytest.mark.parametrize("B,Cin,Cout,Hs", [(2, 16, 8, 6), (1, 128, 64, 16), (3, 32, 96, 4), (1, 6, 6, 1), (2, 64, 32, 32)])
def test_conv3x3_fused_upsample_epilogue(pkg, dev, B, Cin, Cout, Hs):
    """up x2 -> conv -> bias -> noise -> lrelu -> style, one launch, vs the oracle's op chain."""
    tag = f"cvf.{B}.{Cin}.{Cout}.{Hs}"
    x = recipe_input(tag + ".x", (B, Cin, Hs, Hs))
    w = recipe_tensor(tag + ".weight", (Cout, Cin, 3, 3))
    b = recipe_tensor(tag + ".bias", (Cout,))
    nw = recipe_tensor(tag + ".noise.weight", (Cout,))
    nz = recipe_input(tag + ".nz", (B, 1, 2 * Hs, 2 * Hs))
    st = recipe_input(tag + ".style", (B, 2 * Cout))
    ref = F.conv2d(R.upsample2x_bilinear(x), w, b, padding=1)
    ref = F.leaky_relu(R.apply_noise(ref, nw, nz), 0.2)
    ref = ref * (st[:, :Cout].view(B, Cout, 1, 1) + 1.0) + st[:, Cout:].view(B, Cout, 1, 1)
    hosted = 0
    for cfg in range(6):
        if not pkg.ops.conv3x3_config_fits(cfg, B, Cin, Cout, 2 * Hs, 2 * Hs):
            continue
        hosted += 1
        wp = pkg.ops.pack_conv3x3_weight(w.to(dev), cfg)
        for ksplit in (1, 2):
            y = pkg.ops.conv3x3_fused(x.to(dev), wp, Cout, bias=b.to(dev), noise_w=nw.to(dev), noise=nz.to(dev),
                                      style=st.to(dev), upsample=True, lrelu_slope=0.2, config=cfg, ksplit=ksplit)
            assert rel_l2(y, ref) < TOL_OP, (cfg, ksplit)
    assert hosted >= 2


def test_conv3x3_transpose_flip_is_data_gradient(pkg, dev):
    """The transpose_flip packing turns the same kernel into the conv's adjoint (dgrad)."""
    B, Cin, Cout, H = 2, 24, 40, 10
    x = recipe_input("dg.x", (B, Cin, H, H)).requires_grad_(True)
    w = recipe_tensor("dg.weight", (Cout, Cin, 3, 3))
    gy = recipe_input("dg.gy", (B, Cout, H, H))
    F.conv2d(x, w, padding=1).backward(gy)
    cfg = 2
    wp = pkg.ops.pack_conv3x3_weight(w.to(dev), cfg, transpose_flip=True)
    gx = pkg.ops.conv3x3_fused(gy.to(dev), wp, Cin, config=cfg)
    assert rel_l2(gx, x.grad) < TOL_OP


def test_fc_matches_oracle_and_golden(pkg, dev, golden):
    g = golden("decoder_ops.npz")
    for tag, gain, wscale, lrmul, has_bias in [("fc_map", 2 ** 0.5, True, 0.01, True),
                                               ("fc_style", 1.0, True, 1.0, True),
                                               ("fc_plain", 2 ** 0.5, False, 1.0, False)]:
        O, I = g[f"{tag}.gw"].shape
        m = pkg.FC(I, O, gain=gain, use_wscale=wscale, lrmul=lrmul, bias=has_bias)
        m.load_state_dict(fill_state_dict(m.state_dict(), prefix=tag + "."))
        y = m.to(dev)(T(g[f"{tag}.x"]).to(dev))
        assert rel_l2(y, g[f"{tag}.y"]) < TOL_OP, tag
    # batch > 8 (two passes over the weight row) and the 6144-wide first mapping layer
    x = recipe_input("fcbig.x", (11, 6144))
    w = recipe_tensor("fcbig.weight", (96, 6144), 1.0)
    b = recipe_tensor("fcbig.bias", (96,))
    ref = R.fc(x, w, b, 0.02, 0.5)
    y = pkg.ops.fc(x.to(dev), w.to(dev), b.to(dev), 0.02, 0.5, 0.2)
    assert rel_l2(y, ref) < TOL_OP


def test_noise_style_upsample_modules(pkg, dev, golden):
    g = golden("decoder_ops.npz")
    C = g["an.x"].shape[1]
    m = pkg.ApplyNoise(C)
    m.load_state_dict(fill_state_dict(m.state_dict(), prefix="an."))
    assert rel_l2(m.to(dev)(T(g["an.x"]).to(dev), T(g["an.noise"]).to(dev)), g["an.y"]) < TOL_OP
    m = pkg.ApplyStyle(16, C, use_wscale=True)
    m.load_state_dict(fill_state_dict(m.state_dict(), prefix="as."))
    assert rel_l2(m.to(dev)(T(g["as.x"]).to(dev), T(g["as.lat"]).to(dev)), g["as.y"]) < TOL_OP
    for tag in ("up_a", "up_b", "up_c"):
        assert rel_l2(pkg.ops.upsample2x_bilinear(T(g[f"{tag}.x"]).to(dev)), g[f"{tag}.y"]) < TOL_OP


def test_synthesis_block_goldens(pkg, dev, golden):
    g = golden("decoder_blocks.npz")
    for tag, cin, cout, B, hin in [("blk512", 512, 512, 2, 4), ("blk128_64", 128, 64, 1, 16), ("blk16_8", 16, 8, 3, 6)]:
        m = pkg.SynthesisBlock(cin, cout, 3)
        m.load_state_dict(fill_state_dict(m.state_dict(), prefix=tag + "."))
        m.to(dev)
        x = recipe_input(tag + ".x", (B, cin, hin, hin)).to(dev)
        w = recipe_input(tag + ".w", (B, 2, 512)).to(dev)
        n1 = recipe_input(tag + ".n1", (B, 1, 2 * hin, 2 * hin)).to(dev)
        n2 = recipe_input(tag + ".n2", (B, 1, 2 * hin, 2 * hin)).to(dev)
        with torch.no_grad():
            y = m(x, w, n1, n2)
        assert rel_l2(y, g[f"{tag}.y"]) < TOL_OP, tag


def _generator(pkg, dev, prefix="Gd."):
    g = pkg.StyleGenerator(6144).eval()
    sd = fill_state_dict(g.state_dict(), prefix=prefix)
    g.load_state_dict(sd)
    return g.to(dev), sd


def test_style_generator_golden_frame(pkg, dev, golden):
    """BASELINE config 1 input, the reference's own output as the expected value."""
    g, _ = _generator(pkg, dev)
    with torch.no_grad():
        y = g(recipe_input("e2e.features", (1, 6144)).to(dev), [n.to(dev) for n in recipe_noises("e2e", 1, 256)])
    assert y.shape == (1, 3, 256, 256)
    assert rel_l2(y, golden("decoder_e2e_256.npz")["y"]) < TOL_E2E
    gb = golden("decoder_e2e_256_b2.npz")
    with torch.no_grad():
        feats = recipe_input("e2e_b2.features", (2, 6144)).to(dev)
        assert rel_l2(g.mapping(feats), gb["w"]) < TOL_OP
        y = g(feats, [n.to(dev) for n in recipe_noises("e2e_b2", 2, 256)])
    assert rel_l2(y[..., ::4, ::4], gb["y_s4"]) < TOL_E2E
    assert rel_l2(y[..., 96:160, 96:160], gb["y_crop"]) < TOL_E2E


def test_style_generator_batch8_vs_oracle(pkg, dev):
    """BASELINE config 2 (the benchmarked workload): B=8, 256^2, forward."""
    g, sd = _generator(pkg, dev)
    feats = recipe_input("cfg2.features", (8, 6144))
    noises = recipe_noises("cfg2", 8, 256)
    with torch.no_grad():
        y = g(feats.to(dev), [n.to(dev) for n in noises])
        ref = R.style_generator(feats, sd, noises)
    assert y.shape == (8, 3, 256, 256)
    assert rel_l2(y, ref) < TOL_E2E
    # size-independent property: frames are independent -> any sub-batch reproduces its rows exactly
    with torch.no_grad():
        y3 = g(feats[3:4].to(dev), [n[3:4].to(dev) for n in noises])
    # (to summation order: the split-K factor is chosen from the grid size, which depends on the batch)
    assert rel_l2(y3, y[3:4]) < 1e-5


def test_synthesis_512_golden(pkg, dev, golden):
    gold = golden("decoder_e2e_512.npz")
    s = pkg.SynthesisNetwork(resolution=512).eval()
    s.load_state_dict(fill_state_dict(s.state_dict(), prefix="Gd512.synthesis."))
    s.to(dev)
    with torch.no_grad():
        y = s(recipe_input("e2e512.w", (1, 16, 512)).to(dev), [n.to(dev) for n in recipe_noises("e2e512", 1, 512)])
    assert y.shape == (1, 3, 512, 512)
    assert rel_l2(y[..., ::8, ::8], gold["y_s8"]) < TOL_E2E
    assert rel_l2(y[..., 224:288, 224:288], gold["y_crop"]) < TOL_E2E


def test_noise_default_draw_and_errors(pkg, dev):
    """noise=None draws on the device (reference behaviour); CPU tensors are refused loudly."""
    g, _ = _generator(pkg, dev)
    with torch.no_grad():
        y = g(recipe_input("nd.features", (1, 6144)).to(dev))
    assert torch.isfinite(y).all()
    with pytest.raises(RuntimeError):
        pkg.ops.fc(torch.zeros(2, 8), torch.zeros(4, 8))


@pytest.mark.parametrize("B,I,O", [(8, 6144, 512), (3, 2048, 40), (1, 4100, 7), (9, 6144, 64)])
def test_fc_long_rows(pkg, dev, B, I, O):
    """Rows >= 2048 floats take the workgroup-per-row kernel (the 6144 -> 512 first mapping layer)."""
    x, w, b = recipe_input(f"fcw.x.{B}.{I}", (B, I)), recipe_tensor(f"fcw.w.{I}.{O}", (O, I), 1.0), recipe_tensor(f"fcw.b.{O}", (O,), 0.5)
    ref = F.leaky_relu(F.linear(x.double(), w.double() * 0.013, b.double() * 0.7), 0.2)
    y = pkg.ops.fc(x.to(dev), w.to(dev), b.to(dev), 0.013, 0.7, 0.2)
    assert rel_l2(y, ref) < TOL_OP


def test_fc_grouped_equals_separate_launches(pkg, dev):
    """spk_fc_grouped_fwd: the decoder's 13 style affines in one launch -- bitwise equal to 13 spk_fc_fwd calls
    (same wave-per-row arithmetic), and equal to the oracle."""
    torch.manual_seed(4)
    B = 5
    wlat = recipe_input("fcg.w", (B, 14, 512)).to(dev)
    items, refs = [], []
    for j, O in enumerate([1024, 1024, 1024, 512, 512, 256, 128, 130, 64, 6]):
        weight, bias = torch.randn(O, 512, device=dev), torch.randn(O, device=dev)
        items.append((wlat[:, j], weight, bias, 0.0442 + 0.001 * j, 1.0, 0.2))
        refs.append(pkg.ops.fc(wlat[:, j], weight, bias, 0.0442 + 0.001 * j, 1.0, 0.2))
    outs = pkg.ops.fc_grouped(items)
    for o, r, it in zip(outs, refs, items):
        assert torch.equal(o, r)
        cpu = F.leaky_relu(F.linear(it[0].cpu().double(), it[1].cpu().double() * it[3], it[2].cpu().double()), 0.2)
        assert rel_l2(o, cpu) < TOL_OP
    with pytest.raises(pkg._lib.SpkError):
        pkg.ops.fc_grouped(items * 2)          # more than SPK_FC_MAX_GROUPS


def test_hipgraph_replay_matches_eager(pkg, dev):
    """The C ABI neither allocates nor synchronises, so a decoder step can be captured into a hipGraph (what bench.py
    times): the replay is bitwise the eager result, also after the inputs change in place."""
    torch.manual_seed(2)
    g = pkg.StyleGenerator(6144).eval().to(dev)
    with torch.no_grad():
        for n, p in g.named_parameters():
            if "noise" in n:
                p.normal_(0, 0.1)
    B = 2
    feats = torch.randn(B, 6144, device=dev)
    noises = [torch.randn(s, device=dev) for s in g.synthesis.noise_shapes(B)]
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                g(feats, noises)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = g(feats, noises)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, g(feats, noises))
        feats.copy_(torch.randn(B, 6144, device=dev))      # new latents, same buffers
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, g(feats, noises))


def test_launch_plan_equals_launch_by_launch(pkg, dev):
    """The inference forward goes out as one pre-built launch list (plan.DecoderPlan, spk_launch_list); it must be the
    launch-by-launch path bit for bit (same kernels, same arguments) -- with explicit noise, from features and from
    dlatents -- follow in-place parameter updates, and hand back a fresh output tensor per call."""
    g, sd = _generator(pkg, dev)
    B = 3
    feats = recipe_input("plan.features", (B, 6144)).to(dev)
    noises = [n.to(dev) for n in recipe_noises("plan", B, 256)]
    syn = g.synthesis
    PL = importlib.import_module("speak-hack_amd.plan")
    with torch.no_grad():
        # the plan's one extra fusion -- toRGB inside the last Winograd launch's epilogue (SPK_EPI_TORGB) -- sums the 64 channels in
        # another order than the stand-alone 1x1: equal to rounding, checked first; the bit-for-bit part then runs without it
        y_fused = g(feats, noises)
        plan0 = next(iter(g.__dict__["_plans"].values()))
        assert plan0.rgb_fused is not None or not pkg.ops.use_wino(B, 64, 64, 256, 256)
        g.__dict__["_plans"].clear()
        fuse, PL.FUSE_TORGB = PL.FUSE_TORGB, False
    try:
        _plan_equals_launches(pkg, dev, g, syn, feats, noises, y_fused)
    finally:
        PL.FUSE_TORGB = fuse
        g.__dict__["_plans"].clear()
        syn.__dict__["_plans"].clear()


def _plan_equals_launches(pkg, dev, g, syn, feats, noises, y_fused):
    with torch.no_grad():
        y_plan = g(feats, noises)
        assert rel_l2(y_fused, y_plan) < 2e-6 and next(iter(g.__dict__["_plans"].values())).rgb_fused is None
        assert any(k[3] == "features" for k in g.__dict__["_plans"])
        type(syn).use_plan = False
        try:
            y_ref = g(feats, noises)
            w = g.mapping(feats).unsqueeze(1).repeat(1, syn.num_layers, 1).contiguous()
            y_w_ref = syn(w, noises)
        finally:
            type(syn).use_plan = True
        # (the truncation scale rides on the style FCs' multiplier in the plan: one rounding apart from psi * w)
        assert rel_l2(y_plan, y_ref) < 1e-5
        y_w = syn(w, noises)
        assert torch.equal(y_w, y_w_ref)
        assert any(k[3] == "w" for k in syn.__dict__["_plans"])
        # a second call returns a NEW tensor and leaves the first one intact
        keep = y_plan.clone()
        y2 = g(feats * 0.5, noises)
        assert y2.data_ptr() != y_plan.data_ptr() and torch.equal(y_plan, keep) and not torch.equal(y2, keep)
        # in-place parameter update (an optimizer step / load_state_dict): the plan re-packs
        syn.layers[2].conv1.weight.mul_(1.5)
        syn.to_rgb.bias.add_(0.25)
        y3 = g(feats, noises)
        type(syn).use_plan = False
        try:
            y3_ref = g(feats, noises)
        finally:
            type(syn).use_plan = True
        assert rel_l2(y3, y3_ref) < 1e-5 and rel_l2(y3, y_ref) > 1e-3
        # device-drawn noise path: finite, and different draws per call
        a, b = g(feats), g(feats)
        assert torch.isfinite(a).all() and not torch.equal(a, b)


def test_launch_plan_follows_reassigned_parameters_and_truncation(pkg, dev):
    """a plan compared only the Parameter objects it had captured, so ``m.weight = nn.Parameter(...)`` /
    ``load_state_dict(assign=True)`` left eval forwards on the OLD weights; and the truncation scale, baked into the style
    FC multipliers at build time, was not part of the plan key."""
    g, sd = _generator(pkg, dev)
    B = 2
    feats = recipe_input("plan2.features", (B, 6144)).to(dev)
    noises = [n.to(dev) for n in recipe_noises("plan2", B, 256)]
    syn = g.synthesis

    def by_launch():
        type(syn).use_plan = False
        try:
            return g(feats, noises)
        finally:
            type(syn).use_plan = True

    with torch.no_grad():
        y0 = g(feats, noises)
        # ---- a re-assigned Parameter: a NEW object, the old one stays alive inside the plan ----
        conv = syn.layers[1].conv2
        conv.weight = torch.nn.Parameter(conv.weight.detach() * 1.25 + 0.01)
        y1 = g(feats, noises)
        assert rel_l2(y1, y0) > 1e-3                           # the eval output follows the new weight ...
        assert rel_l2(y1, by_launch()) < 1e-5                   # ... and equals the launch-by-launch path
        # ---- load_state_dict(assign=True): every Parameter object replaced ----
        new_sd = {k: (v * 0.9).clone() for k, v in g.state_dict().items()}
        g.load_state_dict(new_sd, assign=True)
        y2 = g(feats, noises)
        assert rel_l2(y2, y1) > 1e-3 and rel_l2(y2, by_launch()) < 1e-5
        # ---- truncation: psi and the cutoff are baked into the plan, so they key it ----
        g.truncation_psi = 0.4
        y3 = g(feats, noises)
        assert rel_l2(y3, y2) > 1e-3 and rel_l2(y3, by_launch()) < 1e-5
        g.truncation_cutoff = 3
        y4 = g(feats, noises)
        assert rel_l2(y4, y3) > 1e-4 and rel_l2(y4, by_launch()) < 1e-5
        g.truncation_psi, g.truncation_cutoff = 0.7, 8
        assert rel_l2(g(feats, noises), y2) < 1e-6
        with pytest.raises(ValueError):
            g(feats, noises[:-1])
    g.load_state_dict(sd)


@pytest.mark.gpu
@pytest.mark.parametrize("train", [False, True])
def test_forward_pair_equals_two_calls(pkg, dev, train):
    """``StyleGenerator.forward_pair`` -- IRFD.forward's two decoder calls (model.py:107-108) as one pass over both batches:
    the same frames and, in training mode, the same host-RNG draws (per-call style-mixing decision and layer), the same
    parameter gradients as two calls in sequence."""
    g = pkg.StyleGenerator(6144)
    g.load_state_dict(fill_state_dict(g.state_dict(), prefix="Gd."))
    g.to(dev).train(train)
    B = 2
    fa, fb = recipe_input("pair.fa", (B, 6144)).to(dev), recipe_input("pair.fb", (B, 6144)).to(dev)
    na = [n.to(dev) for n in recipe_noises("pair.a", B, 256)]
    nb = [n.to(dev) for n in recipe_noises("pair.b", B, 256)]

    def run(pair):
        torch.manual_seed(1234)                   # host draws (mixing decisions) and the device draws of the second latents
        g.zero_grad()
        with torch.set_grad_enabled(train):
            if pair:
                ya, yb = g.forward_pair(fa, fb, na, nb)
            else:
                ya, yb = g(fa, na), g(fb, nb)
            if train:
                ((ya ** 2).mean() + (yb * yb.detach().roll(1, 0)).mean()).backward()
        grads = {k: p.grad.clone() for k, p in g.named_parameters() if p.grad is not None}
        return ya.detach(), yb.detach(), grads

    ya0, yb0, g0 = run(False)
    ya1, yb1, g1 = run(True)
    assert rel_l2(ya1, ya0) < 1e-5 and rel_l2(yb1, yb0) < 1e-5
    assert g0.keys() == g1.keys()
    for k in g0:
        # same products, another summation order (one weight-gradient launch over both batches instead of two and an add); the
        # per-channel noise weights' gradients are cancelling sums over 2^16 - 2^19 pixels and carry that order's rounding
        assert rel_l2(g1[k], g0[k]) < (2e-3 if "noise" in k else 2e-4), k
    # a lone explicit-noise list cannot be paired: the method falls back to two calls
    torch.manual_seed(5)
    with torch.no_grad():
        ya2, yb2 = g.forward_pair(fa, fb, na, None)
    assert ya2.shape == yb2.shape == ya0.shape


def test_weight_gradients_on_the_second_stream_equal_in_order_launches(pkg, dev, monkeypatch):
    """A training pass of the decoder routes its conv weights through ``autograd.WeightGateFn`` and queues their weight gradients
    on ``ops.side_stream``; the gate's backward joins the stream before the parameters see the gradients.  Two calls that share
    the parameters (the gradients of the second are ADDED to the first's by autograd, after both gates), repeated, bitwise equal
    to the in-order schedule."""
    g = pkg.StyleGenerator(6144)
    g.load_state_dict(fill_state_dict(g.state_dict(), prefix="Gd."))
    g.to(dev).train(True)
    g.style_mixing_prob = 0.0
    B = 4
    fa, fb = recipe_input("gate.fa", (B, 6144)).to(dev), recipe_input("gate.fb", (B, 6144)).to(dev)
    na = [n.to(dev) for n in recipe_noises("gate.a", B, 256)]
    nb = [n.to(dev) for n in recipe_noises("gate.b", B, 256)]

    def grads():
        g.zero_grad(set_to_none=True)
        ya, yb = g(fa, na), g(fb, nb)
        ((ya ** 2).mean() + (yb * ya.detach()).mean()).backward()
        return {k: p.grad.clone() for k, p in g.named_parameters() if p.grad is not None}

    assert pkg.ops.side_stream(dev) is not None
    aside = [grads() for _ in range(3)]
    monkeypatch.setattr(pkg.ops, "side_stream", lambda device: None)
    inline = grads()
    for got in aside:
        bad = [k for k in inline if not torch.equal(got[k], inline[k])]
        assert not bad, bad[:4]

