#!/usr/bin/env python3
"""One rank of the data-parallel training ITERATION (train.py:150-210: D step every iteration, G step with the adversarial
term through model.D, clip over all parameters, Adam on Gd) -- ``training.train_iteration`` with both reducers on the HIP
path.  Started by tests/test_dp_gpu.py through ``python -m torch.distributed.run`` (fresh child processes).
``--backend gloo --one-device``: all ranks on GPU 0 (a one-GPU box); ``--backend nccl``: RCCL, one GPU per rank."""
import argparse
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

KW = dict(G_steps=2, r1_weight=1.0, stylegan_loss_weight=0.1, grad_clip_value=1.0)      # config.yaml:18,21,43 (G_steps 5 -> 2: two G steps in 3 iterations)
STEPS = (0, 1, 2)
SEED0 = 1000


def shard(rank, B, dev):
    g = torch.Generator().manual_seed(10 + rank)            # rank r's shard is seed 10 + r
    mk = lambda: (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)
    return {"source_image": mk(), "target_image": mk(), "emotion_labels_s": torch.randint(0, 8, (B,), generator=g).to(dev),
            "emotion_labels_t": torch.randint(0, 8, (B,), generator=g).to(dev)}


def optimizers(net):
    return (torch.optim.Adam(net.Gd.parameters(), lr=2e-4, betas=(0.5, 0.999)),      # config.yaml:19-20
            torch.optim.Adam(net.D.parameters(), lr=5e-5, betas=(0.5, 0.999)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--one-device", action="store_true")
    ap.add_argument("--batch", type=int, default=2, help="samples per rank")
    ap.add_argument("--algo", default="all_reduce")
    args = ap.parse_args()
    rank = int(os.environ["RANK"])
    local = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(args.backend)
    import model as M
    T = importlib.import_module("speak-hack_amd.training")

    torch.manual_seed(1234 + rank)             # replicas are seeded DIFFERENTLY: make_reducers broadcasts rank 0's
    net = M.IRFD().to(dev).train()
    with torch.no_grad():                      # default init zeroes the noise weights: wake that path up
        for n, p in net.named_parameters():
            if ".noise" in n:
                p.normal_(0, 0.1)
    red_G, red_D = T.make_reducers(net, algo=args.algo)
    if rank == 0:
        torch.save(net.state_dict(), os.path.join(args.out, "init.pt"))
    opt_G, opt_D = optimizers(net)
    torch.manual_seed(SEED0 + rank)            # this rank's host + device RNG streams from here on
    batch, log = shard(rank, args.batch, dev), []
    for step in STEPS:
        out = T.train_iteration(net, batch, opt_G, opt_D, step, reducer_G=red_G, reducer_D=red_D, **KW)
        log.append({k: None if v is None else float(v) for k, v in out.items()})
    torch.cuda.synchronize()
    cpu = lambda t: None if t is None else t.detach().cpu().clone()
    torch.save({"params": {k: cpu(v) for k, v in net.named_parameters()}, "buffers": {k: cpu(v) for k, v in net.named_buffers()},
                "grads": {k: cpu(v.grad) for k, v in net.named_parameters()}, "log": log,
                "timeline_D": red_D.timeline(), "timeline_G": red_G.timeline(),
                "buckets": (len(red_G.buckets), len(red_D.buckets))}, os.path.join(args.out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
