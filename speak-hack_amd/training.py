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


def _bce(pred, label):
    return F.binary_cross_entropy_with_logits(pred, torch.full_like(pred, label))


def discriminator_loss(model, batch, *, r1_weight=10.0, real_label=0.9, fake_label=0.1):
    """The forward half of the discriminator step, train.py:159-180: instance-noised real and (detached) reconstructed
    images through ``model.D``, smoothed-label BCE, the R1 penalty on both real images.  Returns (loss_D, r1_reg)."""
    x_s, x_t = batch["source_image"], batch["target_image"]
    loss_D_real = (_bce(model.D(add_instance_noise(x_s)), real_label) + _bce(model.D(add_instance_noise(x_t)), real_label)) / 2
    with torch.no_grad():
        outputs = model(x_s, x_t)
        x_s_recon, x_t_recon = outputs[0], outputs[1]
    loss_D_fake = (_bce(model.D(add_instance_noise(x_s_recon.detach())), fake_label) +
                   _bce(model.D(add_instance_noise(x_t_recon.detach())), fake_label)) / 2
    r1_reg = (compute_r1_reg(model.D, x_s) + compute_r1_reg(model.D, x_t)) / 2
    return loss_D_real + loss_D_fake + r1_weight * r1_reg, r1_reg


def generator_loss(model, batch, *, criterion=reconstruction_criterion, stylegan_loss_weight=1.0, real_label=0.9):
    """The forward half of the generator step, train.py:189-203: ``IRFD.forward`` with grad, the criterion, the
    adversarial term through ``model.D``."""
    x_s, x_t = batch["source_image"], batch["target_image"]
    outputs = model(x_s, x_t)
    l_pose_landmark, l_emotion, l_identity, l_recon = criterion(x_s, x_t, outputs, batch["emotion_labels_s"],
                                                                batch["emotion_labels_t"])
    loss_G_adv = (_bce(model.D(outputs[0]), real_label) + _bce(model.D(outputs[1]), real_label)) / 2
    return l_pose_landmark + l_emotion + l_identity + l_recon + stylegan_loss_weight * loss_G_adv


class IterationAccumulator:
    """``with accelerator.accumulate(model):`` around the loop body (train.py:152; ``gradient_accumulation_steps``,
    train.py:335, config.yaml:27 = 1) as accelerate defines it: ``begin()`` once per iteration; every k-th iteration is a
    synchronising one.  On the others ``accelerator.backward`` still runs (loss / k, no gradient exchange) while the
    wrapped optimizers' ``zero_grad()`` and ``step()`` do nothing.  NOTE the reference calls ``zero_grad()`` at the top
    of each step INSIDE the context, so on the synchronising iteration the gradients piled up by the k-1 before it are
    dropped before its own backward: with k > 1 the reference steps on 1/k of the last micro-batch's gradient.  This
    class reproduces that schedule literally; at k = 1 (the reference's configuration) it is the identity."""

    def __init__(self, steps: int = 1):
        if steps < 1:
            raise ValueError("IterationAccumulator: steps must be >= 1")
        self.steps, self._n, self.sync_gradients = int(steps), 0, True

    def begin(self) -> bool:
        self._n += 1
        self.sync_gradients = self._n % self.steps == 0
        return self.sync_gradients


def train_iteration(model, batch, optimizer_G, optimizer_D, step, *, criterion=reconstruction_criterion, G_steps=5,
                    r1_weight=10.0, stylegan_loss_weight=1.0, grad_clip_value=1.0, real_label=0.9, fake_label=0.1,
                    reducer_G=None, reducer_D=None, accumulator=None):
    """One pass of the loop body of train.py:150-210.  Returns {'loss_D', 'loss_G' | None, 'r1_reg'} as floats-on-device.

    Data parallel (BASELINE config 4; what ``accelerator.prepare`` + DDP would do if the reference ran on more than one
    process): ``batch`` is this rank's shard, ``reducer_D`` = ``dp.GradBucketReducer`` over ``model.D``'s
    parameters, ``reducer_G`` = one over every other parameter of the model.  Both or neither.

    * D step: ``reducer_D`` exchanges during ``loss_D.backward()`` (buckets launched from hooks), ``optimizer_D`` steps on
      the rank-identical means.
    * G step: ``loss_G`` back-propagates THROUGH ``model.D`` (train.py:197-203).  D's hooks stay silent
      (``foreign_backward()``): its local contribution piles up on top of the D step's reduced gradients exactly as in the
      reference, where ``optimizer_D.zero_grad()`` drops it at the next iteration.  ``reducer_G`` exchanges everything
      else while backward runs.  The clip of train.py:207-208 is over ``model.parameters()``, D's included -- a norm over
      rank-local gradients would give every rank its own coefficient and the replicas of ``Gd`` would drift apart -- so
      when clipping is on, D's accumulated gradients are exchanged too (``reducer_D.finish()`` after the backward: 76 MB
      once per ``G_steps`` iterations) and ONE global norm over both reducers' flat buffers scales everything
      (``dp.clip_grad_norm_``).  ``optimizer_G.zero_grad()`` clears ``Gd`` only; encoder / ``Cm`` gradients accumulate
      from one G step to the next as they do in the reference (nobody zeroes or applies them, train.py:346-347).
    * Host and device RNG (instance noise, swap, style mixing, decoder noise) are per rank, as under DDP; BatchNorm
      statistics are per rank (the reference has no SyncBN); spectral-norm u / v stay rank-identical because they depend
      on the weights only.

    ``accumulator``: an ``IterationAccumulator`` (``gradient_accumulation_steps``); None = every iteration synchronises.
    """
    from contextlib import ExitStack
    if (reducer_G is None) != (reducer_D is None):
        raise ValueError("train_iteration: pass both reducer_G and reducer_D, or neither")
    dp = reducer_G is not None
    sync = True if accumulator is None else accumulator.begin()
    scale = 1.0 if accumulator is None or accumulator.steps == 1 else 1.0 / accumulator.steps

    # ---- discriminator step (train.py:155-183) ----
    if sync:
        reducer_D.zero_grad() if dp else optimizer_D.zero_grad()
    loss_D, r1_reg = discriminator_loss(model, batch, r1_weight=r1_weight, real_label=real_label, fake_label=fake_label)
    with ExitStack() as quiet:
        if dp and not sync:
            quiet.enter_context(reducer_D.no_sync())
        (loss_D * scale if scale != 1.0 else loss_D).backward()
    if sync:
        if dp:
            reducer_D.finish()
        optimizer_D.step()
    out = {"loss_D": loss_D.detach(), "r1_reg": r1_reg.detach(), "loss_G": None}

    # ---- generator step, every G_steps iterations (train.py:185-210) ----
    if step % G_steps == 0:
        if sync:
            reducer_G.zero_grad(optimizer_G_params(optimizer_G)) if dp else optimizer_G.zero_grad()
        loss_G = generator_loss(model, batch, criterion=criterion, stylegan_loss_weight=stylegan_loss_weight,
                                real_label=real_label)
        with ExitStack() as quiet:
            if dp:
                quiet.enter_context(reducer_D.foreign_backward())
                if not sync:
                    quiet.enter_context(reducer_G.no_sync())
            (loss_G * scale if scale != 1.0 else loss_G).backward()
        if sync:
            if dp:
                reducer_G.finish()
                if grad_clip_value:
                    reducer_D.finish()                                              # D's share of the global norm
                    _dp().clip_grad_norm_([reducer_G, reducer_D], grad_clip_value)
            elif grad_clip_value:
                torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip_value)     # over ALL parameters (train.py:208)
            optimizer_G.step()
        out["loss_G"] = loss_G.detach()
    return out


def optimizer_G_params(optimizer):
    return [p for g in optimizer.param_groups for p in g["params"]]


def _dp():
    from . import dp
    return dp


def make_reducers(model, **kw):
    """The two reducers ``train_iteration`` takes: (everything but ``model.D``, ``model.D``).  Call once, after the model
    is on its device and torch.distributed is initialised; parameters are broadcast from rank 0, and so are the buffers
    (BatchNorm running statistics, spectral-norm u / v), as DDP's constructor does."""
    import torch.distributed as dist
    dp = _dp()
    d_ids = {id(p) for p in model.D.parameters()}
    reducer_G = dp.GradBucketReducer([p for p in model.parameters() if id(p) not in d_ids], **kw)
    reducer_D = dp.GradBucketReducer(list(model.D.parameters()), **kw)
    if dist.is_initialized() and dist.get_world_size() > 1:
        with torch.no_grad():
            for b in model.buffers():
                dist.broadcast(b, src=0)
    return reducer_G, reducer_D


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
