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
