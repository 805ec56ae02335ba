#!/usr/bin/env python3
"""One rank of the data-parallel IRFD generator step (BASELINE config 4) -- started by tests/test_dp_gpu.py through
``python -m torch.distributed.run`` (fresh child processes; never imported by the test process).

Every rank runs the REAL step on the HIP path: ``IRFD.forward`` (three ResNet-50 encoders with train-mode BatchNorm on
both images, two decoder passes with train-mode style mixing), reconstruction loss, backward, bucketed gradient exchange
through ``dp.GradBucketReducer``, global-norm clip (train.py:186-210).  Phase 1 is the same step on the rank's shard
WITHOUT any exchange (the "single-rank shard gradient" the test averages); phase 2 repeats it through the reducer.
``--backend gloo --one-device`` puts all ranks on GPU 0 (a one-GPU box); ``--backend nccl`` is RCCL, one GPU per rank.
"""
import argparse
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--one-device", action="store_true")
    ap.add_argument("--batch", type=int, default=2, help="samples per rank")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--which", default="g", choices=("g", "d"), help="generator step (train.py:186-210) or discriminator "
                    "step with the R1 double backward (train.py:155-183)")
    ap.add_argument("--accum", type=int, default=1, help="gradient_accumulation_steps (train.py:152,335): the rank's shard "
                    "is split into this many micro-batches, exchanged once through dp.GradAccumulator")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(args.backend)
    import model as M
    import torch.nn.functional as F
    dp = importlib.import_module("speak-hack_amd.dp")
    T = importlib.import_module("speak-hack_amd.training")

    torch.manual_seed(1234 + rank)             # replicas are seeded DIFFERENTLY: the reducer's broadcast makes them equal
    net = M.IRFD().to(dev).train()
    with torch.no_grad():                      # default init zeroes the noise weights: wake that path up
        for n, p in net.named_parameters():
            if ".noise" in n:
                p.normal_(0, 0.1)
    for n, p in net.named_parameters():        # D has its own step (train.py:156-183), with its own exchange
        p.requires_grad_(n.startswith("D.") == (args.which == "d"))
    params = [p for p in net.parameters() if p.requires_grad]
    names = {id(p): n for n, p in net.named_parameters()}
    red = dp.GradBucketReducer(params, bucket_bytes=32 << 20)
    # after the broadcast every rank holds rank 0's weights (BatchNorm buffers are not parameters: copy them too, as
    # DDP's broadcast_buffers does)
    with torch.no_grad():
        for b in net.buffers():
            dist.broadcast(b, src=0)
    buffers0 = {k: v.clone() for k, v in net.named_buffers()}

    B = args.batch
    g = torch.Generator().manual_seed(10 + rank)            # per-rank shard (rank seed 10+r)
    x_s = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)
    x_t = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)

    f_s = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)       # the D step's "reconstructions"
    f_t = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)
    bce = lambda pred, label: F.binary_cross_entropy_with_logits(pred, torch.full_like(pred, label))
    K = args.accum
    assert B % K == 0

    def micro_loss(lo, hi):
        a, b = x_s[lo:hi], x_t[lo:hi]
        if args.which == "d":
            nz = T.add_instance_noise
            return (bce(net.D(nz(a)), 0.9) + bce(net.D(nz(b)), 0.9)) / 2 \
                + (bce(net.D(nz(f_s[lo:hi])), 0.1) + bce(net.D(nz(f_t[lo:hi])), 0.1)) / 2 \
                + 10.0 * (T.compute_r1_reg(net.D, a) + T.compute_r1_reg(net.D, b)) / 2
        out = net(a, b)
        return ((out[0] - a) ** 2).mean() + ((out[1] - b) ** 2).mean()

    def forward_backward(step, acc=None):
        """One optimizer step's worth of backward passes: K micro-batches of B/K samples, each scaled by 1/K; through
        ``acc`` (dp.GradAccumulator) only the last one exchanges."""
        torch.manual_seed(500 + 10 * step + rank)           # host RNG (swap, style-mix draws) and device RNG (noise)
        total = 0.0
        for m in range(K):
            loss = micro_loss(m * B // K, (m + 1) * B // K)
            if acc is not None:
                acc.backward(loss)
            else:
                (loss / K if K > 1 else loss).backward()
            total = total + loss.detach() / K
        return total

    def restore_buffers():
        with torch.no_grad():
            for k, v in net.named_buffers():
                v.copy_(buffers0[k])

    result = {"names": [names[id(p)] for p in params], "steps": {}}
    for step in range(args.steps):
        # ---- phase 1: the shard's own gradients, no exchange ----
        red.zero_grad()
        with red.no_sync():
            loss_local = forward_backward(step)
        local = [None if p.grad is None else p.grad.detach().clone().cpu() for p in params]
        restore_buffers()
        # ---- phase 2: the same step through the reducer ----
        red.zero_grad()
        n_buckets = len(red.buckets)
        cold = [i for i, b in enumerate(red.buckets) if b["cold"]]
        if K > 1:
            acc = dp.GradAccumulator(red, steps=K)
            loss = forward_backward(step, acc)               # the K-th micro-step's backward ends in red.finish()
            assert acc.sync_gradients
            by_hook_before_finish = list(red.stats["launched_by_hook"])
        else:
            loss = forward_backward(step)
            by_hook_before_finish = list(red.stats["launched_by_hook"])
            red.finish()
        total = red.grad_norm()                              # what the global-norm clip uses: rank-identical by construction
        torch.cuda.synchronize()
        restore_buffers()
        result["steps"][step] = {
            "loss_local": float(loss_local), "loss": float(loss), "local": local,
            "reduced": [None if p.grad is None else p.grad.detach().clone().cpu() for p in params],
            "by_hook": by_hook_before_finish, "by_finish": list(red.stats["launched_by_finish"]), "n_buckets": n_buckets,
            "cold": cold, "rebuilt_after": red.rebuilt, "norm": float(total),
            "cold_names": [names[id(p)] for i in cold for p in red.buckets[i]["params"]] if step > 0 else [],
        }
    result["bytes_per_step"] = red.bytes_per_step()
    torch.save(result, os.path.join(args.out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
