#!/usr/bin/env python3
"""Workload for the rocprofv3 counter passes (profiles/): the headline decoder step, launched EAGERLY through its launch
plan (no hipGraph: one dispatch row per kernel), `--steps` times in exact fp32 and `--steps` times on the opt-in bf16x3 path.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_decoder.py
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--sg2", action="store_true", help="the StyleGAN2 variant's step instead (exact fp32 only)")
    args = ap.parse_args()
    pkg = importlib.import_module("speak-hack_amd")
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    if args.sg2:
        sg2 = importlib.import_module("speak-hack_amd.stylegan2")
        gen = sg2.StyleGAN2Generator(6144).eval().to(dev)
        with torch.no_grad():
            for n, p in gen.named_parameters():
                if n.endswith("noise.weight"):
                    p.fill_(0.1)
            z = torch.randn(args.batch, 6144, device=dev)
            for _ in range(args.steps + 1):
                gen(z)
            torch.cuda.synchronize()
        print("done")
        return
    gen = pkg.StyleGenerator(6144).eval().to(dev)
    with torch.no_grad():
        for n, p in gen.named_parameters():
            if "noise" in n:
                p.normal_(0, 0.1)
        feats = torch.randn(args.batch, 6144, device=dev)
        for precision in ("f32", "bf16x3"):
            gen.synthesis.precision = precision
            for _ in range(args.steps + 1):        # the first call of a precision builds its plan (packing kernels)
                gen(feats)
            torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
