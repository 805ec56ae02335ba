"""SURVEY.md 8f F4 on the CPU: the synthetic batch schema (CelebADataset.py:133-138), the state_dict layout a
reference checkpoint has (train.py:236,365) and the checkpoint dict round trip (train.py:235-242,
:364-371).  No kernel is launched: module construction, ``state_dict`` and ``torch.save/load`` only."""
import importlib
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def irfd():
    import model as M
    torch.manual_seed(0)
    return M.IRFD()


def test_synthetic_batch_schema():
    data = importlib.import_module("speak-hack_amd.data")
    ds = data.SyntheticFacePairs(length=6, resolution=32, seed=3)
    a, b = ds[2], ds[2]
    assert set(a) == {"source_image", "target_image", "emotion_labels_s", "emotion_labels_t"}
    assert a["source_image"].shape == (3, 32, 32) and a["source_image"].dtype == torch.float32
    assert float(a["source_image"].min()) >= -1 and float(a["source_image"].max()) <= 1
    assert a["emotion_labels_s"].dtype == torch.long and a["emotion_labels_s"].dim() == 0 and 0 <= int(a["emotion_labels_t"]) < 8
    assert torch.equal(a["target_image"], b["target_image"]) and not torch.equal(ds[1]["source_image"], a["source_image"])
    batch = next(iter(data.synthetic_loader(batch_size=4, length=6, resolution=32)))
    assert batch["source_image"].shape == (4, 3, 32, 32) and batch["emotion_labels_s"].shape == (4,)


def test_state_dict_layout_matches_reference_checkpoints(irfd):
    sd = irfd.state_dict()
    assert {k.split(".")[0] for k in sd} == {"Ei", "Ee", "Ep", "Gd", "D", "Cm"}
    gd = [k for k in sd if k.startswith("Gd.")]
    assert len(gd) == 83 and "Gd.synthesis.layers.5.conv2.weight" in sd and "Gd.mapping.7.bias" in sd
    for e in ("Ei", "Ee", "Ep"):          # torchvision resnet50 children()[:-1] as an nn.Sequential
        ks = [k for k in sd if k.startswith(e + ".")]
        assert len(ks) == 318, (e, len(ks))
        assert f"{e}.0.weight" in sd and sd[f"{e}.0.weight"].shape == (64, 3, 7, 7)
        assert f"{e}.1.num_batches_tracked" in sd and f"{e}.7.2.conv3.weight" in sd and f"{e}.5.0.downsample.1.running_var" in sd
    assert {"D.fromrgb.weight_orig", "D.fromrgb.weight_u", "D.fromrgb.weight_v", "D.fromrgb.bias",
            "D.blocks.0.conv2.weight_orig", "D.dense1.weight_orig"} <= set(sd)
    assert sd["Cm.weight"].shape == (8, 2048)
    n_params = sum(p.numel() for p in irfd.parameters())
    assert 110_000_000 < n_params < 125_000_000          # 3 x 23.5 M + 26.1 M + D


def test_checkpoint_round_trip(irfd, tmp_path):
    T = importlib.import_module("speak-hack_amd.training")
    opt_g = torch.optim.Adam(irfd.Gd.parameters(), lr=1e-4, betas=(0.5, 0.999))
    opt_d = torch.optim.Adam(irfd.D.parameters(), lr=4e-4, betas=(0.5, 0.999))
    path = tmp_path / "best_model-epoch-1-0"
    T.save_checkpoint(path, irfd, opt_g, opt_d, epoch=0, config={"training": {"G_steps": 5}})
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"model_state_dict", "optimizer_G", "optimizer_D", "epoch", "resolution", "config"}   # train.py:235-242
    import model as M
    torch.manual_seed(1)
    other = M.IRFD()
    o_g = torch.optim.Adam(other.Gd.parameters(), lr=1.0)
    o_d = torch.optim.Adam(other.D.parameters(), lr=1.0)
    start, res, cfg = T.load_checkpoint(path, other, o_g, o_d)
    assert (start, res, cfg) == (1, 256, {"training": {"G_steps": 5}})
    assert o_g.param_groups[0]["lr"] == 1e-4 and o_d.param_groups[0]["betas"] == (0.5, 0.999)
    a, b = irfd.state_dict(), other.state_dict()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)


def test_training_iteration_constants_match_reference():
    """The constants of the reference's loop body (train.py:144-149,160-171): label smoothing 0.9 / 0.1 and instance
    noise of std 0.1 on every discriminator input."""
    import inspect
    T = importlib.import_module("speak-hack_amd.training")
    assert inspect.signature(T.add_instance_noise).parameters["std"].default == 0.1
    sig = inspect.signature(T.train_iteration).parameters
    assert sig["real_label"].default == 0.9 and sig["fake_label"].default == 0.1 and sig["G_steps"].default == 5
    torch.manual_seed(0)
    x = torch.zeros(64, 3, 32, 32)
    assert abs(float(T.add_instance_noise(x).std()) - 0.1) < 5e-3


def test_train_conv_precision_switch_is_scoped():
    """``ops.train_conv_precision`` (the opt-in reduced-precision training switch) restores the previous setting on exit and on
    error, rejects unknown names, and never routes anything while it is off."""
    import importlib
    ops = importlib.import_module("speak-hack_amd").ops
    assert ops.TRAIN_CONV_PRECISION == "f32"
    with ops.train_conv_precision("bf16x3"):
        assert ops.TRAIN_CONV_PRECISION == "bf16x3"
        with ops.train_conv_precision("f32"):
            assert ops.TRAIN_CONV_PRECISION == "f32"
        assert ops.TRAIN_CONV_PRECISION == "bf16x3"
    assert ops.TRAIN_CONV_PRECISION == "f32"
    try:
        with ops.train_conv_precision("bf16x3"):
            raise RuntimeError("boom")
    except RuntimeError:
        pass
    assert ops.TRAIN_CONV_PRECISION == "f32"
    import pytest
    with pytest.raises(ValueError):
        with ops.train_conv_precision("fp8"):
            pass
    assert not ops.train_bf16x3(8, 64, 64, 256, 256)          # off: nothing takes the split-precision kernel
