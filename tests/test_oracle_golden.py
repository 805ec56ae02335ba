"""Pins the CPU oracle (oracle/*.py) against the golden vectors produced by the reference's own
code (tools/make_goldens.py).  CPU only.  Tolerance: 2e-6 rel-L2 in fp32 (the oracle and the
reference run the same ATen kernels in a possibly different association order)."""
import numpy as np
import torch

from conftest import rel_l2
from oracle import decoder_ref as R
from oracle import legacy_ops_ref as LG
from oracle.weights_recipe import fill_state_dict, recipe_input, recipe_noises, recipe_tensor

TOL = 2e-6
torch.set_num_threads(8)


def T(a):
    return torch.from_numpy(np.asarray(a))


def decoder_template_sd(resolution=256, input_dim=6144, with_mapping=True, prefix=""):
    """Key -> shape layout of the decoder state dict, independent of any module."""
    import math
    sd = {}
    if with_mapping:
        for i in range(8):
            sd[f"mapping.{i}.weight"] = (512, input_dim if i == 0 else 512)
            sd[f"mapping.{i}.bias"] = (512,)
    sd["synthesis.const_input"] = (1, 512, 4, 4)
    sd["synthesis.bias"] = (512,)
    sd["synthesis.style_mod.linear.weight"] = (1024, 512)
    sd["synthesis.style_mod.linear.bias"] = (1024,)
    sd["synthesis.noise_input1.weight"] = (512,)
    layers = R.decoder_conv_layers(resolution)
    for i in range(int(math.log2(resolution)) - 2):
        cin, cout = layers[2 * i][0], layers[2 * i][1]
        p = f"synthesis.layers.{i}."
        sd[p + "conv1.weight"] = (cout, cin, 3, 3)
        sd[p + "conv1.bias"] = (cout,)
        sd[p + "conv2.weight"] = (cout, cout, 3, 3)
        sd[p + "conv2.bias"] = (cout,)
        sd[p + "noise1.weight"] = (cout,)
        sd[p + "noise2.weight"] = (cout,)
        for s in ("style_mod1", "style_mod2"):
            sd[p + s + ".linear.weight"] = (2 * cout, 512)
            sd[p + s + ".linear.bias"] = (2 * cout,)
    sd["synthesis.to_rgb.weight"] = (3, layers[-1][1], 1, 1)
    sd["synthesis.to_rgb.bias"] = (3,)
    return {k: recipe_tensor(prefix + k, v) for k, v in sd.items()}


def test_fc_variants(golden):
    g = golden("decoder_ops.npz")
    for tag, gain, wscale, lrmul, has_bias in [("fc_map", 2 ** 0.5, True, 0.01, True),
                                               ("fc_style", 1.0, True, 1.0, True),
                                               ("fc_plain", 2 ** 0.5, False, 1.0, False)]:
        x = T(g[f"{tag}.x"]).requires_grad_(True)
        O, I = g[f"{tag}.gw"].shape
        w = recipe_tensor(f"{tag}.weight", (O, I)).requires_grad_(True)
        b = recipe_tensor(f"{tag}.bias", (O,)).requires_grad_(True) if has_bias else None
        wl, bl = R.wscale_fc(I, gain, wscale, lrmul)
        y = R.fc(x, w, b, wl, bl)
        assert rel_l2(y, g[f"{tag}.y"]) < TOL
        y.backward(T(g[f"{tag}.gy"]))
        assert rel_l2(x.grad, g[f"{tag}.gx"]) < TOL
        assert rel_l2(w.grad, g[f"{tag}.gw"]) < TOL
        if has_bias:
            assert rel_l2(b.grad, g[f"{tag}.gb"]) < TOL


def test_noise_style_upsample(golden):
    g = golden("decoder_ops.npz")
    x = T(g["an.x"]).requires_grad_(True)
    w = recipe_tensor("an.weight", (x.shape[1],)).requires_grad_(True)
    y = R.apply_noise(x, w, T(g["an.noise"]))
    assert rel_l2(y, g["an.y"]) < TOL
    y.backward(T(g["an.gy"]))
    assert rel_l2(x.grad, g["an.gx"]) < TOL and rel_l2(w.grad, g["an.gw"]) < TOL

    x = T(g["as.x"]).requires_grad_(True)
    lat = T(g["as.lat"]).requires_grad_(True)
    C = x.shape[1]
    lw = recipe_tensor("as.linear.weight", (2 * C, 16)).requires_grad_(True)
    lb = recipe_tensor("as.linear.bias", (2 * C,)).requires_grad_(True)
    y = R.apply_style(x, lat, lw, lb)
    assert rel_l2(y, g["as.y"]) < TOL
    y.backward(T(g["as.gy"]))
    for a, k in [(x.grad, "as.gx"), (lat.grad, "as.glat"), (lw.grad, "as.gw"), (lb.grad, "as.gb")]:
        assert rel_l2(a, g[k]) < TOL

    for tag in ("up_a", "up_b", "up_c"):
        x = T(g[f"{tag}.x"]).requires_grad_(True)
        y = R.upsample2x_bilinear(x)
        assert rel_l2(y, g[f"{tag}.y"]) < TOL
        y.backward(T(g[f"{tag}.gy"]))
        assert rel_l2(x.grad, g[f"{tag}.gx"]) < TOL


def test_synthesis_blocks_fwd_bwd(golden):
    g = golden("decoder_blocks.npz")
    for tag, cin, cout, B, hin in [("blk512", 512, 512, 2, 4), ("blk128_64", 128, 64, 1, 16), ("blk16_8", 16, 8, 3, 6)]:
        shapes = {"conv1.weight": (cout, cin, 3, 3), "conv1.bias": (cout,), "conv2.weight": (cout, cout, 3, 3),
                  "conv2.bias": (cout,), "noise1.weight": (cout,), "noise2.weight": (cout,),
                  "style_mod1.linear.weight": (2 * cout, 512), "style_mod1.linear.bias": (2 * cout,),
                  "style_mod2.linear.weight": (2 * cout, 512), "style_mod2.linear.bias": (2 * cout,)}
        sd = {k: recipe_tensor(f"{tag}.{k}", s).requires_grad_(True) for k, s in shapes.items()}
        x = recipe_input(tag + ".x", (B, cin, hin, hin)).requires_grad_(True)
        w = recipe_input(tag + ".w", (B, 2, 512)).requires_grad_(True)
        n1 = recipe_input(tag + ".n1", (B, 1, 2 * hin, 2 * hin))
        n2 = recipe_input(tag + ".n2", (B, 1, 2 * hin, 2 * hin))
        y = R.synthesis_block(x, w, sd, "", n1, n2)
        assert rel_l2(y, g[f"{tag}.y"]) < TOL, tag
        y.backward(recipe_input(tag + ".gy", y.shape))
        assert rel_l2(x.grad, g[f"{tag}.gx"]) < 5e-6
        assert rel_l2(w.grad, g[f"{tag}.gw"]) < 5e-6
        for k, p in sd.items():
            if f"{tag}.g.{k}" in g:
                assert rel_l2(p.grad, g[f"{tag}.g.{k}"]) < 5e-6, (tag, k)
            else:
                sl = p.grad[:8, :8] if p.grad.dim() == 4 else p.grad[:8, :64]
                assert rel_l2(sl, g[f"{tag}.g.{k}.slice"]) < 5e-6, (tag, k)
                assert abs(float(p.grad.double().norm()) / float(g[f"{tag}.g.{k}.norm"]) - 1) < 1e-5


def test_style_generator_eval_full_frame(golden):
    """BASELINE config 1: single 256^2 frame, CPU reference."""
    g = golden("decoder_e2e_256.npz")
    sd = decoder_template_sd(prefix="Gd.")
    with torch.no_grad():
        y = R.style_generator(recipe_input("e2e.features", (1, 6144)), sd, recipe_noises("e2e", 1, 256))
    assert y.shape == (1, 3, 256, 256)
    assert rel_l2(y, g["y"]) < TOL

    g = golden("decoder_e2e_256_b2.npz")
    feats = recipe_input("e2e_b2.features", (2, 6144))
    with torch.no_grad():
        assert rel_l2(R.mapping(feats, sd), g["w"]) < TOL
        y = R.style_generator(feats, sd, recipe_noises("e2e_b2", 2, 256))
    assert rel_l2(y[..., ::4, ::4], g["y_s4"]) < TOL
    assert rel_l2(y[..., 96:160, 96:160], g["y_crop"]) < TOL


def test_style_generator_train_mixing_and_grads(golden):
    g = golden("decoder_train_mix.npz")
    sd = {k: v.requires_grad_(True) for k, v in decoder_template_sd(prefix="Gd.").items()}
    feats = recipe_input("mix.features", (1, 6144)).requires_grad_(True)
    y = R.style_generator(feats, sd, recipe_noises("mix", 1, 256), mix_features=T(g["mix_features"]),
                          mix_layer=int(g["mix_layer"]))
    assert rel_l2(y[..., ::4, ::4], g["y_s4"]) < TOL
    y.backward(recipe_input("mix.gy", y.shape))
    checks = [(feats.grad, "gfeat"), (sd["mapping.7.bias"].grad, "g_map7_bias"),
              (sd["synthesis.const_input"].grad, "g_const"), (sd["synthesis.to_rgb.weight"].grad, "g_rgb_w"),
              (sd["synthesis.to_rgb.bias"].grad, "g_rgb_b"),
              (sd["synthesis.layers.5.noise2.weight"].grad, "g_l5_noise2"),
              (sd["synthesis.layers.5.conv2.bias"].grad, "g_l5_conv2_b"),
              (sd["synthesis.layers.5.conv2.weight"].grad, "g_l5_conv2_w"),
              (sd["synthesis.layers.0.conv1.weight"].grad[:8, :8], "g_l0_conv1_w_slice"),
              (sd["synthesis.layers.3.style_mod1.linear.bias"].grad, "g_l3_style1_b")]
    for a, k in checks:
        assert rel_l2(a, g[k]) < 2e-5, k


def test_synthesis_512(golden):
    """BASELINE config 5 decoder: SynthesisNetwork(resolution=512), 16 w rows."""
    g = golden("decoder_e2e_512.npz")
    sd = decoder_template_sd(resolution=512, with_mapping=False, prefix="Gd512.")
    assert R.synthesis_num_layers(512) == 16
    w = recipe_input("e2e512.w", (1, 16, 512))
    with torch.no_grad():
        y = R.synthesis_network(w, sd, recipe_noises("e2e512", 1, 512), resolution=512)
    assert y.shape == (1, 3, 512, 512)
    assert rel_l2(y[..., ::8, ::8], g["y_s8"]) < TOL
    assert rel_l2(y[..., 224:288, 224:288], g["y_crop"]) < TOL


def test_legacy_ops(golden):
    g = golden("legacy_ops.npz")
    fns = {"pixelnorm": LG.pixel_norm, "instnorm": LG.instance_norm, "blur": LG.blur2d,
           "blur_s2": lambda x: LG.blur2d(x, stride=2), "blur_flip": lambda x: LG.blur2d(x, f=(1, 2, 3), flip=True),
           "upscale": LG.upscale2d, "upscale_g": lambda x: LG.upscale2d(x, 2, 0.5),
           "pixelnorm_sqrt": LG.pixel_norm_sqrt}
    for tag, fn in fns.items():
        x = T(g["x"]).requires_grad_(True)
        y = fn(x)
        assert rel_l2(y, g[f"{tag}.y"]) < TOL, tag
        y.backward(T(g[f"{tag}.gy"]))
        assert rel_l2(x.grad, g[f"{tag}.gx"]) < TOL, tag


def test_fused_upscale(golden):
    """styleganv1.py:231: the ConvTranspose2d(4, s2, p1) of the reference's own GBlock(res=7) (512 -> 256 channels)."""
    from oracle.weights_recipe import recipe_tensor
    g = golden("legacy_fused_upscale.npz")
    w = recipe_tensor("legacy.fused_upscale.weight", (512, 256, 4, 4), 1.0) * (512 * 4) ** -0.5
    b = recipe_tensor("legacy.fused_upscale.bias", (256,), 0.5)
    assert rel_l2(LG.fused_upscale(T(g["x"]), w, b), g["y"]) < TOL
    # round 3: the reference module's own backward (gx, gb in full, gw sampled + its norm)
    from oracle.weights_recipe import recipe_input
    x = T(g["x"]).clone().requires_grad_(True)
    w, b = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    LG.fused_upscale(x, w, b).backward(recipe_input("legacy.fused_upscale.gy", tuple(g["y"].shape)))
    assert rel_l2(x.grad, g["gx"]) < TOL and rel_l2(b.grad, g["gb"]) < TOL
    assert rel_l2(w.grad[::8, ::4], g["gw_sample"]) < TOL and abs(float(w.grad.double().norm()) / float(g["gw_norm"]) - 1) < 1e-5


def test_flop_accounting_matches_survey():
    f = R.decoder_flops_per_frame(256)
    assert abs(f["total"] / 1e9 - 56.214) < 0.01
    assert abs(f["conv3x3"] / 1e9 - 56.17) < 0.01
