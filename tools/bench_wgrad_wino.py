#!/usr/bin/env python3
"""Winograd weight gradient (csrc/wgrad3x3_wino_f32.hip) against the direct kernel, per decoder / discriminator layer shape:
relative L2 error of both against an fp64 convolution's weight gradient, and time (kernel + slab reduce).

    python tools/bench_wgrad_wino.py [--batch 8] [--splits 0]
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd")
if os.environ.get("SPK_LAB_LIB"):
    pkg._lib.LIB_PATH = os.path.abspath(os.environ["SPK_LAB_LIB"])
ops = pkg.ops


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def ref64(g, x):
    x64 = x.double().requires_grad_(False)
    w = torch.zeros(g.shape[1], x.shape[1], 3, 3, device=g.device, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x64, w, padding=1)
    (dw,) = torch.autograd.grad(y, w, g.double())
    return dw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--wgs", type=int, default=0, help="choose the splits so that the grid has about this many workgroups")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()
    dev, B = torch.device("cuda:0"), args.batch
    torch.manual_seed(0)
    shapes = [(512, 512, 16), (512, 512, 32), (512, 256, 64), (256, 256, 64), (256, 128, 128), (128, 128, 128), (128, 64, 256), (64, 64, 256),
              (64, 128, 256), (128, 256, 128), (256, 512, 64), (512, 512, 32)]
    tot_w = tot_d = 0.0
    for ci, co, r in shapes:
        g = torch.randn(B, co, r, r, device=dev)
        x = torch.randn(B, ci, r, r, device=dev)
        if not ops.wgrad_wino_supported(B, ci, co, r, r):
            print(f"{ci:4d}->{co:4d} {r:3d}^2: not served")
            continue
        if args.wgs:
            args.splits = max(1, args.wgs // ((ci // 64) * (co // 64)))
        sp = pkg._lib.lib().spk_conv2d_wgrad_wino_splits(args.splits, B, ci, co, r, r)
        dw_w = ops.conv2d_wgrad_wino(g, x, co, ci, splits=args.splits)
        with ops.conv3x3_algo("direct"):
            dw_d = ops.conv2d_wgrad(g, x, co, ci, 3, 1)
        torch.cuda.synchronize()
        err = ""
        if not args.no_check:
            ref = ref64(g, x)
            ew = ((dw_w.double() - ref).norm() / ref.norm()).item()
            ed = ((dw_d.double() - ref).norm() / ref.norm()).item()
            err = f"  rel-L2 wino {ew:.2e} direct {ed:.2e}"
            del ref
        tw = timed(lambda: ops.conv2d_wgrad_wino(g, x, co, ci, splits=args.splits))
        with ops.conv3x3_algo("direct"):
            td = timed(lambda: ops.conv2d_wgrad(g, x, co, ci, 3, 1))
        fl = 2 * 9 * ci * co * r * r * B
        tot_w += tw
        tot_d += td
        print(f"{ci:4d}->{co:4d} {r:3d}^2 splits {sp:4d}: wino {tw * 1e3:8.1f} us ({fl / tw / 1e9:6.1f} alg TF/s)  direct {td * 1e3:8.1f} us "
              f"({fl / td / 1e9:6.1f})  x{td / tw:.2f}{err}", flush=True)
    print(f"total wino {tot_w:.2f} ms, direct {tot_d:.2f} ms")


if __name__ == "__main__":
    main()
