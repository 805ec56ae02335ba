"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/spk.h declares (no compute calls -- there is no GPU here); the
host-side module mirrors keep the reference's state_dict layout; CPU tensors are refused."""
import ctypes
import importlib
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    ge.build()
    return importlib.import_module("speak-hack_amd")


def header_symbols():
    src = open(os.path.join(ROOT, "include", "spk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    lib = ctypes.CDLL(pkg._lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/spk.h but not exported"
    assert sorted(pkg._lib.exported_symbols()) == syms, "ctypes prototypes out of sync with include/spk.h"
    assert b"gfx950" in pkg._lib.lib().spk_version()


def test_host_side_queries_and_argument_errors(pkg):
    lib = pkg._lib.lib()
    n = lib.spk_conv2d_num_configs()
    assert n >= 4
    for cfg in range(n):
        co, ci, px = pkg.ops.conv3x3_config_info(cfg)
        if lib.spk_conv2d_config_valid(cfg, 7, 7, 2) and not lib.spk_conv2d_config_valid(cfg, 3, 3, 2):
            # the 7x7 stride-2 stem form (Cin 3 -> 64 per group): packed [148 k][64 co], k = (ci, ky, kx), one zero row
            assert (co, ci, px) == (64, 3, 512)
            assert lib.spk_conv2d_packed_floats(cfg, 7, 7, 3, 64) == 148 * 64 and lib.spk_conv2d_packed_floats(cfg, 7, 7, 4, 64) == -1
            assert lib.spk_conv2d_packed_floats(cfg, 3, 3, 3, 64) == -1
            assert lib.spk_conv2d_pick_config(7, 7, 2, 2, 3, 64, 32, 32) == cfg and lib.spk_conv2d_pick_config(7, 7, 2, 2, 3, 64, 32, 30) != cfg
            assert lib.spk_conv2d_stats_slots(cfg, 7, 7, 2, 2, 3, 64, 40, 36) == 2 * 3 * 2      # 16 x 32 pixel tiles per image
            continue
        assert co % 32 == 0 and px % 32 == 0 and ci % 2 == 0
        if lib.spk_conv2d_config_valid(cfg, 1, 1, 1) and not lib.spk_conv2d_config_valid(cfg, 1, 1, 2):
            # the GEMM forms of a stride-1 1x1: the packed image is the plain [Cout][Cin] matrix (32-channel k-tiles) or
            # [co tile][k tile][16][CO_T], zero padded (16-channel k-tiles)
            want = 40 if ci == 32 else -(-5 // co) * -(-8 // 16) * 16 * co
            assert lib.spk_conv2d_packed_floats(cfg, 1, 1, 8, 5) == want and lib.spk_conv2d_packed_floats(cfg, 3, 3, 3, 5) == -1
            continue
        if lib.spk_conv2d_config_valid(cfg, 2, 2, 1) and not lib.spk_conv2d_config_valid(cfg, 3, 3, 1):
            # the exact-tap data gradient of a 3x3 stride-2 conv: the transposed 3x3 operator in 64 x 8 channel tiles
            assert (co, ci) == (64, 8) and lib.spk_conv2d_packed_floats(cfg, 2, 2, 3, 4 * 128) == 2 * 1 * 9 * 8 * 64
            assert lib.spk_conv2d_dgrad_s2_config(2, 3, 128, 8, 16) == cfg and lib.spk_conv2d_dgrad_s2_config(2, 3, 128, 8, 8) < 4
            assert lib.spk_conv2d_packed_floats(cfg, 3, 3, 3, 5) == -1
            continue
        # packed image is zero-padded up to whole tiles
        assert lib.spk_conv2d_packed_floats(cfg, 3, 3, 3, 5) == -(-5 // co) * -(-3 // ci) * 9 * ci * co
    assert lib.spk_conv2d_packed_floats(99, 3, 3, 3, 5) < 0
    # error behaviour: negative code + message, never an exception/abort from C
    assert lib.spk_conv2d_fwd(None, None) < 0
    assert b"null" in lib.spk_last_error()
    assert lib.spk_fc_fwd(None, 0, None, None, None, 0, 1, 1, 1, 1.0, 1.0, 1.0, None) < 0


def test_decoder_state_dict_layout_matches_reference(pkg):
    """Key layout and shapes of SURVEY.md 8(b) (83 tensors, 26,076,867 parameters)."""
    g = pkg.StyleGenerator(6144)
    sd = g.state_dict()
    assert len(sd) == 83
    assert sum(p.numel() for p in g.parameters()) == 26076867
    assert tuple(sd["mapping.0.weight"].shape) == (512, 6144)
    assert tuple(sd["synthesis.const_input"].shape) == (1, 512, 4, 4)
    assert tuple(sd["synthesis.style_mod.linear.weight"].shape) == (1024, 512)
    assert tuple(sd["synthesis.layers.3.conv1.weight"].shape) == (256, 512, 3, 3)
    assert tuple(sd["synthesis.layers.5.style_mod2.linear.bias"].shape) == (128,)
    assert tuple(sd["synthesis.to_rgb.weight"].shape) == (3, 64, 1, 1)
    assert g.input_dim == 6144 and g.synthesis.num_layers == 14
    assert pkg.SynthesisNetwork(resolution=512).num_layers == 16
    # the top-level drop-in module exposes the reference's names
    import styleganv1
    assert styleganv1.StyleGenerator is pkg.StyleGenerator


def test_no_cpu_fallback(pkg):
    g = pkg.StyleGenerator(6144).eval()
    with pytest.raises(RuntimeError, match="HIP|device|CPU"):
        with torch.no_grad():
            g(torch.zeros(1, 6144))


def test_product_path_never_imports_oracle():
    """The shipped package must not reach into oracle/ (test infrastructure)."""
    pk = os.path.join(ROOT, "speak-hack_amd")
    for dirpath, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
    for f in ("styleganv1.py", "model.py", "stylegan.py"):
        assert "oracle" not in open(os.path.join(ROOT, f)).read()


def test_winograd_shape_queries_on_the_host(pkg):
    """The Winograd entry points' host-side answers (no launch): which shapes the conv / the weight gradient serve, the packed
    image size, the split rule of the weight gradient's one-round grid and its workspace."""
    lib = pkg._lib.lib()
    # forward: regions of 32 x 8 output pixels (16 x 16 where the image is narrower), input channels in chunks of 8 paired two by two
    assert lib.spk_conv2d_wino_supported(8, 64, 64, 256, 256) and lib.spk_conv2d_wino_supported(1, 16, 200, 8, 32)
    assert lib.spk_conv2d_wino_supported(8, 64, 64, 16, 16) and lib.spk_conv2d_wino_supported(1, 64, 64, 32, 48)
    assert not lib.spk_conv2d_wino_supported(8, 64, 64, 8, 8) and not lib.spk_conv2d_wino_supported(8, 64, 64, 8, 16)
    assert not lib.spk_conv2d_wino_supported(8, 64, 64, 12, 32)       # H % 8
    # the sliced contraction: few (region, channel tile) pairs -> the smallest power of two that fills 3/4 of the CUs, >= 4 chunks a slice
    ks = lib.spk_conv2d_wino_ksplit
    assert ks(0, 8, 512, 512, 16, 16) == 4 and ks(0, 8, 512, 512, 32, 32) == 1 and ks(0, 1, 512, 512, 32, 32) == 8
    assert ks(0, 1, 64, 64, 16, 16) == 2 and ks(0, 1, 16, 64, 16, 16) == 1           # (8 / 2 chunks: at most 2 / 1 slices)
    assert ks(3, 8, 512, 512, 16, 16) == 2 and ks(16, 8, 512, 512, 16, 16) == 16 and ks(1, 8, 512, 512, 16, 16) == 1
    assert ks(0, 8, 64, 64, 8, 8) == -1
    wsb = lib.spk_conv2d_wino_workspace_bytes
    assert wsb(0, 8, 512, 512, 16, 16) == 4 * 8 * 512 * 256 * 4 and wsb(0, 8, 512, 512, 32, 32) == 0 and wsb(0, 8, 64, 64, 8, 8) == -1
    assert not lib.spk_conv2d_wino_supported(8, 24, 64, 32, 32)       # Cin % 16
    assert not lib.spk_conv2d_wino_supported(64, 512, 512, 256, 256)  # >= 2 GB: 32-bit gather offsets
    assert lib.spk_conv2d_packed_bytes_wino(64, 64) == 64 * 64 * 16 * 4 and lib.spk_conv2d_packed_bytes_wino(64, 65) == 64 * 128 * 16 * 4
    # weight gradient: 64 x 64 (co, ci) blocks, chunks of 16 x 2 output pixels
    sup = lib.spk_conv2d_wgrad_wino_supported
    assert sup(8, 64, 64, 256, 256) and sup(1, 64, 128, 2, 16) and sup(3, 192, 64, 6, 48)
    assert not sup(8, 3, 64, 256, 256) and not sup(8, 64, 96, 32, 32) and not sup(8, 64, 64, 32, 24) and not sup(8, 64, 64, 7, 32)
    assert not sup(64, 512, 64, 256, 256)
    for (B, ci, co, H, W) in [(8, 64, 64, 256, 256), (8, 512, 512, 32, 32), (8, 512, 512, 16, 16), (3, 64, 64, 6, 48), (1, 64, 64, 2, 16)]:
        chunks, blocks = B * (H // 2) * (W // 16), (ci // 64) * (co // 64)
        s = lib.spk_conv2d_wgrad_wino_splits(0, B, ci, co, H, W)
        assert 1 <= s <= max(1, 256 // blocks) and s <= (chunks + 1) // 2
        per = 2 * -(-chunks // (2 * s))
        assert (s - 1) * per < chunks <= s * per                      # every workgroup has at least one real chunk
        assert s < 8 or s % 8 == 0 or s * per >= chunks               # (a multiple of 8 where that many are asked for)
        assert lib.spk_conv2d_wgrad_wino_workspace_bytes(0, B, ci, co, H, W) == s * co * 9 * ci * 4
        s3 = lib.spk_conv2d_wgrad_wino_splits(3, B, ci, co, H, W)       # an explicit request is honoured up to the chunk count
        assert 1 <= s3 <= 3
    assert lib.spk_conv2d_wgrad_wino_splits(0, 8, 64, 96, 32, 32) == -1 and lib.spk_conv2d_wgrad_wino_workspace_bytes(0, 8, 64, 96, 32, 32) == -1
