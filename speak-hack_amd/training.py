"""The reference's training iteration and checkpoint format on the HIP path (train.py:150-242,
:246-255, :364-371).  ``train.py`` itself cannot run here (accelerate / omegaconf / dlib / hsemotion / network weight
fetches); this module restates its per-iteration schedule -- discriminator step with instance noise, real/fake BCE
and the R1 penalty, then every ``G_steps`` iterations the generator step with the adversarial term, global-norm
clipping over ALL model parameters and Adam on ``Gd`` -- around a pluggable criterion, because ``IRFDLoss`` needs
third-party networks that are not available offline (out of scope).  Everything that touches activations
runs through ``IRFD`` / ``StyleDiscriminator`` and therefore on the HIP kernels; this file only sequences them.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def add_instance_noise(x, std=0.1):
    """train.py:148-149 (std 0.1 on all four real / fake discriminator inputs, :160-171)."""
    return x + torch.randn_like(x) * std


def compute_r1_reg(D, real_img):
    """train.py:246-255 verbatim in behaviour: mean over the batch of |dD/dx|^2, differentiable w.r.t. D's weights."""
    from . import autograd as AG
    real_img = real_img.detach().requires_grad_(True)
    real_pred = D(real_img)
    with AG.input_grad_only():      # dD/dx only: skip the per-layer weight-gradient kernels autograd would discard
        (grad_real,) = torch.autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
    return grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()


def reconstruction_criterion(x_s, x_t, outputs, emotion_labels_s, emotion_labels_t):
    """Stand-in for ``IRFDLoss`` (model.py:128-240, needs dlib / hsemotion / 6DRepNet): the parts computable from the
    model's own outputs -- L2 reconstruction (model.py ``l_recon``) and the emotion cross-entropy on ``Cm``'s softmax
    outputs.  Returns the same 4-tuple order (pose_landmark, emotion, identity, recon) with zeros for the others."""
    x_s_recon, x_t_recon = outputs[0], outputs[1]
    l_recon = F.mse_loss(x_s_recon, x_s) + F.mse_loss(x_t_recon, x_t)
    l_emotion = F.nll_loss(torch.log(outputs[8] + 1e-8), emotion_labels_s) + F.nll_loss(torch.log(outputs[9] + 1e-8), emotion_labels_t)
    zero = l_recon.new_zeros(())
    return zero, l_emotion, zero, l_recon


def train_iteration(model, batch, optimizer_G, optimizer_D, step, *, criterion=reconstruction_criterion, G_steps=5,
                    r1_weight=10.0, stylegan_loss_weight=1.0, grad_clip_value=1.0, real_label=0.9, fake_label=0.1):
    """One pass of the loop body of train.py:150-210.  Returns {'loss_D', 'loss_G' | None, 'r1_reg'} as floats-on-device."""
    x_s, x_t = batch["source_image"], batch["target_image"]
    bce = lambda pred, label: F.binary_cross_entropy_with_logits(pred, torch.full_like(pred, label))

    # ---- discriminator step (train.py:155-183) ----
    optimizer_D.zero_grad()
    loss_D_real = (bce(model.D(add_instance_noise(x_s)), real_label) + bce(model.D(add_instance_noise(x_t)), real_label)) / 2
    with torch.no_grad():
        outputs = model(x_s, x_t)
        x_s_recon, x_t_recon = outputs[0], outputs[1]
    loss_D_fake = (bce(model.D(add_instance_noise(x_s_recon.detach())), fake_label) +
                   bce(model.D(add_instance_noise(x_t_recon.detach())), fake_label)) / 2
    r1_reg = (compute_r1_reg(model.D, x_s) + compute_r1_reg(model.D, x_t)) / 2
    loss_D = loss_D_real + loss_D_fake + r1_weight * r1_reg
    loss_D.backward()
    optimizer_D.step()
    out = {"loss_D": loss_D.detach(), "r1_reg": r1_reg.detach(), "loss_G": None}

    # ---- generator step, every G_steps iterations (train.py:185-210) ----
    if step % G_steps == 0:
        optimizer_G.zero_grad()
        outputs = model(x_s, x_t)
        l_pose_landmark, l_emotion, l_identity, l_recon = criterion(x_s, x_t, outputs, batch["emotion_labels_s"],
                                                                    batch["emotion_labels_t"])
        loss_G_adv = (bce(model.D(outputs[0]), real_label) + bce(model.D(outputs[1]), real_label)) / 2
        loss_G = l_pose_landmark + l_emotion + l_identity + l_recon + stylegan_loss_weight * loss_G_adv
        loss_G.backward()
        if grad_clip_value:
            torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip_value)     # over ALL parameters (train.py:208)
        optimizer_G.step()
        out["loss_G"] = loss_G.detach()
    return out


def save_checkpoint(path, model, optimizer_G, optimizer_D, epoch, config=None):
    """The dict of train.py:235-242 (same keys; ``accelerator.save`` is ``torch.save`` on the main process)."""
    torch.save({"model_state_dict": model.state_dict(), "optimizer_G": optimizer_G.state_dict(),
                "optimizer_D": optimizer_D.state_dict(), "epoch": epoch, "resolution": model.current_resolution,
                "config": config}, path)


def load_checkpoint(path, model, optimizer_G=None, optimizer_D=None, map_location=None):
    """train.py:364-371.  Returns (start_epoch, resolution, config)."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(ck["model_state_dict"])
    if optimizer_G is not None:
        optimizer_G.load_state_dict(ck["optimizer_G"])
    if optimizer_D is not None:
        optimizer_D.load_state_dict(ck["optimizer_D"])
    return ck["epoch"] + 1, ck["resolution"], ck.get("config")
