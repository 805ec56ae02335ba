"""Multi-process (world size 2, gloo, CPU) test of the data-parallel gradient exchange
(speak-hack_amd/dp.py): bucketed, hook-driven all-reduce with grads living inside the buckets.
The reducer is backend-agnostic plumbing; on the GPU box the same code runs over RCCL."""
import importlib
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _toy():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(12, 40), nn.Tanh(), nn.Linear(40, 40), nn.Tanh(), nn.Linear(40, 3))


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = importlib.import_module("speak-hack_amd.dp")
    model = _toy()
    unused = nn.Parameter(torch.ones(5))                       # never receives a gradient
    params = list(model.parameters()) + [unused]
    if rank == 1:                                              # replicas need not be seeded alike: rank 0 is broadcast
        with torch.no_grad():
            for p in params:
                p.add_(1.0)
    red = dp.GradBucketReducer(params, bucket_bytes=4096)      # tiny buckets -> several collectives
    assert len(red.buckets) >= 3
    assert torch.equal(model[0].weight, _toy()[0].weight) and torch.equal(unused, torch.ones(5))
    g = torch.Generator().manual_seed(100)
    x_all, y_all = torch.randn(8, 12, generator=g), torch.randn(8, 3, generator=g)
    lo, hi = dp.shard_batch(8, rank, world)
    out = {}
    for step in range(3):                                      # later steps check zero_grad / re-arm / the rebuilt buckets
        red.zero_grad()
        loss = ((model(x_all[lo:hi]) - y_all[lo:hi]) ** 2).mean()
        loss.backward()
        hook_launched, n_buckets = list(red.stats["launched_by_hook"]), len(red.buckets)
        red.finish()                                           # (the first finish() also re-buckets)
        out[step] = [p.grad.clone() for p in params]
        assert all(p.grad.data_ptr() >= b["flat"].data_ptr() for b in red.buckets for p in b["params"])
        assert red.rebuilt
        launched = red.stats["launched_by_hook"] + red.stats["launched_by_finish"]
        assert launched == list(range(n_buckets))              # strictly in index order, every bucket exactly once
        if step >= 1:
            # buckets are in autograd-completion order now: everything but the never-used parameter's own
            # trailing bucket went out from a hook, i.e. while backward was still running
            cold = [i for i, b in enumerate(red.buckets) if b["cold"]]
            assert cold == [len(red.buckets) - 1] and [id(p) for p in red.buckets[-1]["params"]] == [id(unused)]
            assert hook_launched == list(range(len(red.buckets) - 1)), (hook_launched, len(red.buckets))
    # gradient accumulation (accelerator.accumulate, train.py:152): two micro-batches per exchange
    acc = dp.GradAccumulator(red, steps=2)
    red.zero_grad()
    mid = (lo + hi) // 2
    for a, b in ((lo, mid), (mid, hi)):
        synced = acc.backward(((model(x_all[a:b]) - y_all[a:b]) ** 2).mean())
    assert synced and acc.sync_gradients
    out["accum"] = [p.grad.clone() for p in params]
    norm = red.grad_norm()
    red.clip_(0.01)
    out["norm"], out["clipped"] = norm, [p.grad.clone() for p in params]
    torch.save(out, os.path.join(tmp, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_full_batch_gradients(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    # reference: mean over ranks of per-shard gradients == what DDP leaves in .grad
    model = _toy()
    g = torch.Generator().manual_seed(100)
    x_all, y_all = torch.randn(8, 12, generator=g), torch.randn(8, 3, generator=g)
    ref = None
    for lo, hi in ((0, 4), (4, 8)):
        model.zero_grad()
        ((model(x_all[lo:hi]) - y_all[lo:hi]) ** 2).mean().backward()
        gs = [p.grad.clone() for p in model.parameters()]
        ref = gs if ref is None else [a + b for a, b in zip(ref, gs)]
    ref = [t / world for t in ref] + [torch.zeros(5)]
    for step in (0, 1, 2, "accum"):                              # two half micro-batches of MSE means == the shard's mean
        for a, b, c in zip(r0[step], r1[step], ref):
            assert torch.allclose(a, b, atol=0, rtol=0)            # ranks hold identical gradients
            assert torch.allclose(a, c, atol=1e-6, rtol=1e-5)
    total = torch.sqrt(sum((t.double() ** 2).sum() for t in ref)).float()
    assert torch.allclose(r0["norm"], total, rtol=1e-5)
    coef = min(1.0, 0.01 / (float(total) + 1e-6))
    for a, c in zip(r0["clipped"], ref):
        assert torch.allclose(a, c * coef, atol=1e-7, rtol=1e-4)


def _gan_worker(rank, world, port, tmp):
    """The D-step / G-step interleaving of train.py:155-210 with one reducer per network: the generator loss
    backpropagates THROUGH the discriminator, whose reducer must stay silent during that backward."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = importlib.import_module("speak-hack_amd.dp")
    torch.manual_seed(3)
    G = nn.Sequential(nn.Linear(6, 24), nn.Tanh(), nn.Linear(24, 12))
    D = nn.Sequential(nn.Linear(12, 24), nn.Tanh(), nn.Linear(24, 1))
    red_G = dp.GradBucketReducer(G.parameters(), bucket_bytes=512)
    red_D = dp.GradBucketReducer(D.parameters(), bucket_bytes=512)
    calls = {"n": 0}
    real_all_reduce = dist.all_reduce

    def counting(*a, **k):
        calls["n"] += 1
        return real_all_reduce(*a, **k)
    dist.all_reduce = counting
    g = torch.Generator().manual_seed(7)
    z_all, x_all = torch.randn(8, 6, generator=g), torch.randn(8, 12, generator=g)
    lo, hi = dp.shard_batch(8, rank, world)
    z, x = z_all[lo:hi], x_all[lo:hi]
    out = {}
    for it in range(2):
        # ---- D step: its own reducer exchanges ----
        red_D.zero_grad()
        loss_D = D(x).mean() - D(G(z).detach()).mean()
        loss_D.backward()
        red_D.finish()
        out[("D", it)] = [p.grad.clone() for p in D.parameters()]
        # ---- G step: backward runs through D; D's reducer is muted, G's exchanges ----
        red_G.zero_grad()
        before = calls["n"]
        with red_D.foreign_backward():
            (-D(G(z)).mean()).backward()
        launched_in_backward = calls["n"] - before
        n_G = len(red_G.buckets)
        red_G.finish()
        assert calls["n"] - before == n_G, (calls["n"] - before, n_G)      # exactly G's buckets, none of D's
        if it == 1:
            assert launched_in_backward == n_G                             # (rebuilt order: all from hooks)
        out[("G", it)] = [p.grad.clone() for p in G.parameters()]
        # D's gradients: the reduced D-step values plus THIS rank's local G-step contribution (the reference accumulates
        # them too and optimizer_D.zero_grad() drops them, train.py:157)
        out[("D_after_G", it)] = [p.grad.clone() for p in D.parameters()]
    # a synchronising backward that is never finish()ed: zero_grad() must wait for its collectives, and the next step
    # must come out right (handles were dropped un-waited)
    red_D.zero_grad()
    (D(x).mean() - D(G(z).detach()).mean()).backward()
    assert red_D._handles                                                   # launched from hooks, not waited for
    red_D.zero_grad()
    assert not red_D._handles and all(p.grad is None for p in D.parameters())
    (D(x).mean() - D(G(z).detach()).mean()).backward()
    red_D.finish()
    out["D_again"] = [p.grad.clone() for p in D.parameters()]
    dist.all_reduce = real_all_reduce
    torch.save(out, os.path.join(tmp, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_generator_backward_through_the_discriminator_keeps_its_reducer_silent(tmp_path):
    world = 2
    mp.spawn(_gan_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    torch.manual_seed(3)
    G = nn.Sequential(nn.Linear(6, 24), nn.Tanh(), nn.Linear(24, 12))
    D = nn.Sequential(nn.Linear(12, 24), nn.Tanh(), nn.Linear(24, 1))
    g = torch.Generator().manual_seed(7)
    z_all, x_all = torch.randn(8, 6, generator=g), torch.randn(8, 12, generator=g)
    # full-batch gradients == mean over ranks of the shard gradients (means over equal shards)
    D.zero_grad()
    (D(x_all).mean() - D(G(z_all).detach()).mean()).backward()
    ref_D = [p.grad.clone() for p in D.parameters()]
    G.zero_grad(); D.zero_grad()
    (-D(G(z_all)).mean()).backward()
    ref_G = [p.grad.clone() for p in G.parameters()]
    local_D = []
    for lo, hi in ((0, 4), (4, 8)):
        D.zero_grad()
        (-D(G(z_all[lo:hi])).mean()).backward()
        local_D.append([p.grad.clone() for p in D.parameters()])
    for it in (0, 1):
        for a, b, c in zip(r0[("D", it)], r1[("D", it)], ref_D):
            assert torch.equal(a, b) and torch.allclose(a, c, atol=1e-6, rtol=1e-5)
        for a, b, c in zip(r0[("G", it)], r1[("G", it)], ref_G):
            assert torch.equal(a, b) and torch.allclose(a, c, atol=1e-6, rtol=1e-5)
        for r, loc in ((r0, local_D[0]), (r1, local_D[1])):
            for a, d, l in zip(r[("D_after_G", it)], ref_D, loc):
                assert torch.allclose(a, d + l, atol=1e-6, rtol=1e-5)
    for a, b, c in zip(r0["D_again"], r1["D_again"], ref_D):
        assert torch.equal(a, b) and torch.allclose(a, c, atol=1e-6, rtol=1e-5)


def test_single_process_reducer_is_a_noop_exchange():
    dp = importlib.import_module("speak-hack_amd.dp")
    model = _toy()
    red = dp.GradBucketReducer(model.parameters(), bucket_bytes=1 << 20)
    red.zero_grad()
    x = torch.randn(4, 12)
    model(x).sum().backward()
    red.finish()
    ref = _toy()
    ref(x).sum().backward()
    for p, q in zip(model.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad)
    assert all(b["flat"] is None for b in red.buckets)         # one rank: nothing to exchange, nothing is copied
    total = torch.sqrt(sum((q.grad.double() ** 2).sum() for q in ref.parameters())).float()
    assert torch.allclose(red.grad_norm(), total, rtol=1e-5)
    red.clip_(0.5 * float(total))
    for p, q in zip(model.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad * 0.5, rtol=1e-4, atol=1e-7)
    red.zero_grad()
    assert all(p.grad is None for p in model.parameters())     # set_to_none: the next backward's gradients are adopted
    acc = dp.GradAccumulator(red, steps=3)                     # one rank: accumulation still follows accelerate's schedule
    flags = [acc.backward(model(x).sum()) for _ in range(3)]
    assert flags == [False, False, True]
    for p, q in zip(model.parameters(), ref.parameters()):     # 3 x (loss / 3)
        assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-6)
    red.zero_grad()
    assert red.bytes_per_step() == sum(p.numel() * 4 for p in model.parameters())
    assert dp.shard_batch(64, 3, 8) == (24, 32)
