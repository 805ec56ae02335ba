"""Single-process emulation of an N-rank data-parallel run of ``training.train_iteration`` -- test infrastructure.

What the data-parallel iteration must equal is "the single-process iteration on the concatenated batch", i.e. the step on
the MEAN over shards of the per-shard gradients.  A literal concatenated batch cannot be compared number for number:
BatchNorm statistics are per rank (the reference has no SyncBN), and each rank draws its own instance noise, swap, style
mixing and decoder noise, exactly as under DDP.  So the emulation plays the ranks one after the other on ONE model:
``as_rank(r)`` swaps in rank r's host + device RNG state and its copy of the model's buffers (BatchNorm running
statistics, spectral-norm u / v) and swaps them out again afterwards; parameters and gradients are shared, each shard's
loss is scaled by 1/N, so ``param.grad`` ends up holding the mean -- what the exchange leaves on every rank.
"""
import contextlib
import importlib

import torch


class RankStates:
    def __init__(self, model, seeds, device=None):
        self.model, self.device = model, device
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.rng, self.buffers = [], []
        keep_cpu = torch.get_rng_state()
        keep_dev = torch.cuda.get_rng_state(device) if self.cuda else None
        for seed in seeds:
            torch.manual_seed(seed)                     # seeds the host generator and every device generator
            self.rng.append((torch.get_rng_state(), torch.cuda.get_rng_state(device) if self.cuda else None))
            self.buffers.append({k: v.detach().clone() for k, v in model.named_buffers()})
        torch.set_rng_state(keep_cpu)
        if self.cuda:
            torch.cuda.set_rng_state(keep_dev, device)

    @contextlib.contextmanager
    def as_rank(self, r):
        cpu, dev = self.rng[r]
        torch.set_rng_state(cpu)
        if self.cuda:
            torch.cuda.set_rng_state(dev, self.device)
        with torch.no_grad():
            for k, v in self.model.named_buffers():
                v.copy_(self.buffers[r][k])
        try:
            yield
        finally:
            self.rng[r] = (torch.get_rng_state(), torch.cuda.get_rng_state(self.device) if self.cuda else None)
            self.buffers[r] = {k: v.detach().clone() for k, v in self.model.named_buffers()}


def emulate(model, shards, optimizer_G, optimizer_D, steps, seeds, device=None, **kw):
    """``steps`` iterations of train.py:150-210 over ``len(shards)`` emulated ranks.  ``kw``: the keyword arguments of
    ``training.train_iteration`` (G_steps, r1_weight, ...).  Returns the per-iteration losses of every rank."""
    T = importlib.import_module("speak-hack_amd.training")
    world = len(shards)
    states = RankStates(model, seeds, device)
    G_steps = kw.get("G_steps", 5)
    clip = kw.get("grad_clip_value", 1.0)
    d_kw = {k: kw[k] for k in ("r1_weight", "real_label", "fake_label") if k in kw}
    g_kw = {k: kw[k] for k in ("criterion", "stylegan_loss_weight", "real_label") if k in kw}
    log = []
    for step in steps:
        rec = {"loss_D": [], "loss_G": []}
        optimizer_D.zero_grad()
        for r in range(world):
            with states.as_rank(r):
                loss_D, _ = T.discriminator_loss(model, shards[r], **d_kw)
                (loss_D / world).backward()
                rec["loss_D"].append(float(loss_D.detach()))
        optimizer_D.step()
        if step % G_steps == 0:
            optimizer_G.zero_grad()
            for r in range(world):
                with states.as_rank(r):
                    loss_G = T.generator_loss(model, shards[r], **g_kw)
                    (loss_G / world).backward()
                    rec["loss_G"].append(float(loss_G.detach()))
            if clip:
                torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
            optimizer_G.step()
        log.append(rec)
    return log, states
