"""The ResNet-50 trunk oracle (oracle/resnet_ref.py) is "parity unpinned" by the reference (the
arithmetic lives in un-vendored torchvision).  It is pinned here against an independent
implementation of the same published architecture: transformers' ResNetModel built from config
(random init, no download), weights mapped key by key.  CPU only."""
import pytest
import torch

from conftest import rel_l2
from oracle import resnet_ref as RR
from oracle.weights_recipe import recipe_input, resnet_trunk_state_dict

torch.set_num_threads(8)


def hf_model_from(sd):
    tr = pytest.importorskip("transformers")
    from transformers import ResNetConfig, ResNetModel
    m = ResNetModel(ResNetConfig())      # depths (3,4,6,3), hidden (256,512,1024,2048), bottleneck, v1.5 stride placement
    hf = m.state_dict()
    mapped = {}

    def put(dst, src):
        assert dst in hf and tuple(hf[dst].shape) == tuple(sd[src].shape), (dst, src)
        mapped[dst] = sd[src]

    def put_bn(dst, src):
        for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            put(f"{dst}.{k}", f"{src}.{k}")

    put("embedder.embedder.convolution.weight", "0.weight")
    put_bn("embedder.embedder.normalization", "1")
    for li, nblk in enumerate(RR.LAYERS):
        for bi in range(nblk):
            h, t = f"encoder.stages.{li}.layers.{bi}", f"{4 + li}.{bi}"
            for j in range(3):
                put(f"{h}.layer.{j}.convolution.weight", f"{t}.conv{j + 1}.weight")
                put_bn(f"{h}.layer.{j}.normalization", f"{t}.bn{j + 1}")
            if bi == 0:
                put(f"{h}.shortcut.convolution.weight", f"{t}.downsample.0.weight")
                put_bn(f"{h}.shortcut.normalization", f"{t}.downsample.1")
    assert set(mapped) == set(hf), set(hf) - set(mapped)
    m.load_state_dict(mapped)
    return m


def test_layout_and_flops():
    shapes = RR.trunk_param_shapes()
    n_params = sum(int(torch.tensor(s).prod()) if s else 1 for k, s in shapes.items()
                   if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_params == 23508032
    assert abs(RR.trunk_flops(256, 256) / 1e9 - 10.677) < 0.12   # SURVEY.md 2a (hook count incl. small terms)


def test_trunk_matches_independent_implementation_eval_and_train():
    sd = resnet_trunk_state_dict("Ei.")
    m = hf_model_from(sd)
    x = recipe_input("resnet.x", (2, 3, 96, 96), "uniform")
    m.eval()
    with torch.no_grad():
        ref = m(x).pooler_output
        out = RR.resnet50_trunk(x, sd)
    assert out.shape == (2, 2048, 1, 1)
    assert rel_l2(out, ref) < 1e-5
    # training-mode BatchNorm (batch statistics), running statistics updated in place when asked
    m.train()
    sd_t = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        ref = m(x).pooler_output
        out = RR.resnet50_trunk(x, sd_t, training=True, update_running_stats=True)
    assert rel_l2(out, ref) < 1e-5
    hf = m.state_dict()
    assert rel_l2(sd_t["1.running_mean"], hf["embedder.embedder.normalization.running_mean"]) < 1e-5
    assert rel_l2(sd_t["7.2.bn3.running_var"], hf["encoder.stages.3.layers.2.layer.2.normalization.running_var"]) < 1e-5
