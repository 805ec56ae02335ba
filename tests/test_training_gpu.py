"""SURVEY.md 8f F4 on the GPU: two iterations of the reference's training schedule (train.py:150-210: D step with
instance noise + R1 every iteration, G step with the adversarial term, clip over all parameters and Adam on Gd every
``G_steps``) driven by the synthetic loader, all activations on the HIP path; then the checkpoint round trip."""
import importlib
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_two_training_iterations_and_checkpoint(tmp_path):
    assert torch.cuda.is_available()
    import model as M
    T = importlib.import_module("speak-hack_amd.training")
    data = importlib.import_module("speak-hack_amd.data")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = M.IRFD().to(dev).train()
    opt_g = torch.optim.Adam(net.Gd.parameters(), lr=1e-4, betas=(0.5, 0.999))
    opt_d = torch.optim.Adam(net.D.parameters(), lr=4e-4, betas=(0.5, 0.999))
    loader = data.synthetic_loader(batch_size=2, length=4, resolution=256, seed=5)
    snap = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items() if v.dtype.is_floating_point}
    for step, batch in enumerate(loader):
        batch = {k: v.to(dev) for k, v in batch.items()}
        before_g, before_d, before_e = snap(net.Gd), snap(net.D), net.Ei[0].weight.detach().clone()
        out = T.train_iteration(net, batch, opt_g, opt_d, step, G_steps=2)
        assert torch.isfinite(out["loss_D"]) and torch.isfinite(out["r1_reg"]) and float(out["r1_reg"]) >= 0
        after_g, after_d = snap(net.Gd), snap(net.D)
        assert any(not torch.equal(before_d[k], after_d[k]) for k in before_d if k.endswith("weight_orig"))   # Adam on D
        if step % 2 == 0:
            assert out["loss_G"] is not None and torch.isfinite(out["loss_G"])
            assert not torch.equal(before_g["synthesis.to_rgb.weight"], after_g["synthesis.to_rgb.weight"])
            assert not torch.equal(before_g["mapping.0.weight"], after_g["mapping.0.weight"])
            assert net.Ei[0].weight.grad is not None          # encoders receive (clipped) gradients ...
        else:
            assert out["loss_G"] is None
            assert all(torch.equal(before_g[k], after_g[k]) for k in before_g)
        assert torch.equal(before_e, net.Ei[0].weight.detach())   # ... but only Gd is stepped (train.py:346)
    path = tmp_path / "best_model-epoch-1-1"
    T.save_checkpoint(path, net, opt_g, opt_d, epoch=0)
    torch.manual_seed(9)
    other = M.IRFD().to(dev)
    o_g, o_d = torch.optim.Adam(other.Gd.parameters()), torch.optim.Adam(other.D.parameters())
    assert T.load_checkpoint(path, other, o_g, o_d, map_location=dev)[0] == 1
    a, b = net.state_dict(), other.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert len(o_g.state) == len(opt_g.state) > 0 and len(o_d.state) == len(opt_d.state) > 0
