#!/usr/bin/env python3
"""BASELINE configs 3 / 4: the IRFD generator step -- 3 ResNet-50 encoders (train-mode BatchNorm,
checkpoint recompute) x 2 images + StyleGAN decoder x 2, forward + backward of
mean((x_s_recon-x_s)^2) + mean((x_t_recon-x_t)^2), global-norm clip over all parameters
(train.py:207-208) and Adam on Gd (train.py:346,210) -- per GPU batch B, data parallel over ranks with
the bucketed gradient all-reduce of speak-hack_amd/dp.py.

    python tools/train_step_bench.py --batch 8 --steps 5                       # 1 GPU
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_step_bench.py --batch 8

Prints one JSON line from rank 0: pairs/s (a pair = source+target image), ms/step, algorithmic TFLOP/s
(593.5 GFLOP per pair) and the gradient bytes exchanged per step.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--decoder-only", action="store_true")
    ap.add_argument("--sg2-decoder", action="store_true", help="swap IRFD.Gd for the build-defined StyleGAN2 variant "
                    "(BASELINE config 3 read literally)")
    ap.add_argument("--d-step", action="store_true", help="the discriminator step of train.py:155-183 instead "
                    "(4 D passes with BCE + 2 R1 penalties, backward, Adam on D); fakes are synthetic images")
    ap.add_argument("--precision", default="f32", choices=("f32", "bf16x3"), help="bf16x3: the opt-in reduced-precision training "
                    "switch (ops.train_conv_precision): 3x3 stride-1 convs forward + data gradients on the bf16 pipe")
    ap.add_argument("--graph", action="store_true", help="EXPERIMENT: capture one step (fixed host-RNG decisions) as a hipGraph "
                    "and time its replays -- what the launch gaps of the eager step are worth")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    import model as M
    dp = importlib.import_module("speak-hack_amd.dp")
    torch.manual_seed(0)                       # identical initial weights on every rank
    net = M.IRFD()
    if args.sg2_decoder:
        net.Gd = importlib.import_module("speak-hack_amd.stylegan2").StyleGAN2Generator(6144)
    net = net.to(dev).train()
    if args.d_step:
        for n, p in net.named_parameters():
            p.requires_grad_(n.startswith("D."))
    else:
        for p in net.D.parameters():           # D has its own step (train.py:156-183); not part of the G step
            p.requires_grad_(False)
    params = [p for p in net.parameters() if p.requires_grad]
    red = dp.GradBucketReducer(params)
    opt = torch.optim.Adam(net.D.parameters() if args.d_step else net.Gd.parameters(), lr=1e-4, betas=(0.5, 0.999),
                           capturable=args.graph)
    torch.manual_seed(10 + rank)               # per-rank data and per-rank host RNG
    B = args.batch
    x_s = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
    x_t = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1

    import torch.nn.functional as F
    fake_s = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
    fake_t = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1

    T = importlib.import_module("speak-hack_amd.training")

    def r1(x):                                 # train.py:246-255
        return T.compute_r1_reg(net.D, x)

    def bce(pred, label):
        return F.binary_cross_entropy_with_logits(pred, torch.full_like(pred, label))

    def d_step():
        red.zero_grad()
        noise = T.add_instance_noise               # std 0.1, train.py:148-149
        loss = (bce(net.D(noise(x_s)), 0.9) + bce(net.D(noise(x_t)), 0.9)) / 2 \
            + (bce(net.D(noise(fake_s)), 0.1) + bce(net.D(noise(fake_t)), 0.1)) / 2 + 10.0 * (r1(x_s) + r1(x_t)) / 2
        loss.backward()
        red.finish()
        opt.step()
        return loss

    def step():
        if args.d_step:
            return d_step()
        red.zero_grad()
        if args.decoder_only:
            f = torch.randn(B, 6144, device=dev)
            loss = ((net.Gd(f) - x_s) ** 2).mean() + ((net.Gd(f) - x_t) ** 2).mean()
        else:
            out = net(x_s, x_t)
            loss = ((out[0] - x_s) ** 2).mean() + ((out[1] - x_t) ** 2).mean()
        loss.backward()
        red.finish()
        red.clip_(1.0)
        opt.step()
        return loss

    ops = importlib.import_module("speak-hack_amd").ops
    ops.TRAIN_CONV_PRECISION = args.precision          # (the context manager's global, set for the whole run)
    graph = None
    if args.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(args.warmup, 3)):
                step()
            with torch.no_grad():                       # every packed-weight cache keys on the version: the captured step repacks
                for p in net.parameters():
                    p.add_(0)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = step()
        torch.cuda.synchronize()
        graph.replay()
    else:
        if os.environ.get("SPK_MAIN_PRIO"):               # LAB: the whole step on a stream of this priority (-1 = high)
            print("stream priority range", torch.cuda.Stream.priority_range(), file=sys.stderr)
            main = torch.cuda.Stream(priority=int(os.environ["SPK_MAIN_PRIO"]))
            main.wait_stream(torch.cuda.current_stream())
            torch.cuda.set_stream(main)
        for _ in range(args.warmup):
            step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            loss = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank == 0:
        gflop_pair = (2 * 3 * 56.214) if args.decoder_only else 593.5
        pairs = world * B * args.steps / el
        print(json.dumps({"metric": "discriminator-step pairs/s (4 D fwd+bwd + 2 R1 double backward + Adam)" if args.d_step
                          else "IRFD generator-step pairs/s (fwd+bwd+clip+Adam)", "value": round(pairs, 2),
                          "unit": "pairs/s", "n_gpus": world, "batch_per_gpu": B, "ms_per_step": round(el / args.steps * 1e3, 2),
                          "algorithmic_tflops": None if args.d_step else round(pairs * gflop_pair / 1e3, 1),
                          "dtype": "f32" if args.precision == "f32" else "f32 with bf16 hi+lo conv operands (opt-in)", "scaling": "weak",
                          "grad_bytes_per_step": red.bytes_per_step(), "buckets": len(red.buckets),
                          "loss": round(float(loss), 5), "decoder_only": args.decoder_only}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
