#!/usr/bin/env python3
"""Decoder training-step timing on the GPU box: StyleGenerator fwd + bwd of mean((G(z)-x)^2), B per flag.

    python tools/bench_train.py [--batch 8] [--steps 10]

Prints ms/step for forward-only, forward+backward, and the per-kernel split if run under rocprofv3.
"""
import argparse
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--sg2", action="store_true", help="the build-defined StyleGAN2 variant instead of the StyleGAN1-style decoder")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    if args.sg2:
        gen = importlib.import_module("speak-hack_amd.stylegan2").StyleGAN2Generator(6144).train().to(dev)
    else:
        gen = pkg.StyleGenerator(6144).train().to(dev)
    with torch.no_grad():
        for n, p in gen.named_parameters():
            if "noise" in n:
                p.normal_(0, 0.1)
    B = args.batch
    feats = torch.randn(B, 6144, device=dev)
    target = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1

    def fwd():
        with torch.no_grad():
            return gen(feats)

    def fwdbwd():
        gen.zero_grad(set_to_none=True)
        y = gen(feats)
        loss = ((y - target) ** 2).mean()
        loss.backward()
        return loss

    for name, fn in (("forward", fwd), ("forward+backward", fwdbwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        flops = 56.214e9 * B * (1 if name == "forward" else 3)
        print(f"{name:18s} B={B}: {ms:8.2f} ms/step  {B / ms * 1e3:8.1f} frames/s  ~{flops / ms / 1e9:6.1f} TFLOP/s (algorithmic)")


if __name__ == "__main__":
    main()
