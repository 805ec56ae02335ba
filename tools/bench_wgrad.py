#!/usr/bin/env python3
"""Per-layer timing of the weight-gradient kernel on the decoder's 3x3 convs (B = 8): kernel + slab reduce.

    python tools/bench_wgrad.py [--batch 8] [--res 256]
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd")
if os.environ.get("SPK_LAB_LIB"):          # a lab build of the library (knock-out variants), e.g. tools/_bin/libspk_hip_lab.so
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


def stride2(dev, B):
    tot_t = tot_f = 0.0
    rows = [("D", 64, 128, 128, 1), ("D", 128, 256, 64, 1), ("D", 256, 512, 32, 1), ("D", 512, 512, 16, 1), ("D", 512, 512, 8, 1),
            ("trunk", 128, 128, 32, 6), ("trunk", 256, 256, 16, 6), ("trunk", 512, 512, 8, 6)]
    for name, ci, co, r, G in rows:        # r = OUTPUT size
        g = torch.randn(B, G * co, r, r, device=dev)
        x = torch.randn(B, G * ci, 2 * r, 2 * r, device=dev)
        kw = {}
        if G > 1:
            kw = dict(groups=G, in_affine=(torch.rand(G * ci, device=dev) + 0.5, torch.randn(G * ci, device=dev) * 0.1))
        ms = timed(lambda: ops.conv2d_wgrad(g, x, co, ci, 3, 2, **kw))
        fl = 2 * 9 * ci * co * r * r * B * G
        tot_t += ms
        tot_f += fl
        print(f"{name:5s} {ci:4d}->{co:4d} out {r:3d}^2 x{G}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TFLOP/s")
    print(f"total {tot_t:.2f} ms, {tot_f / tot_t / 1e9:.1f} TFLOP/s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--upsample", action="store_true", help="the first conv of every block reads the bilinear x2 of its input: "
                    "time the folded form against upsample-then-wgrad")
    ap.add_argument("--stride2", action="store_true", help="the 3x3 stride-2 layers instead: the discriminator's conv2 of every block "
                    "and the trunk's three downsampling convs (6 groups, BatchNorm-folded input)")
    args = ap.parse_args()
    dev, B = torch.device("cuda:0"), args.batch
    if args.stride2:
        return stride2(dev, B)
    r, cin, tot_t, tot_f = 8, 512, 0.0, 0.0
    while r <= args.res:
        cout = min(int(8192 / (2.0 ** (r.bit_length() - 2))), 512)
        for ci, co in ((cin, cout), (cout, cout)):
            g = torch.randn(B, co, r, r, device=dev)
            x = torch.randn(B, ci, r, r, device=dev)
            for splits in (0,):
                for _ in range(2):
                    ops.conv2d_wgrad(g, x, co, ci, 3, 1, splits=splits)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.conv2d_wgrad(g, x, co, ci, 3, 1, splits=splits)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 5
                fl = 2 * 9 * ci * co * r * r * B
                tot_t += ms
                tot_f += fl
                print(f"{ci:4d}->{co:4d} @{r:3d}^2 splits={splits}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TFLOP/s")
        if args.upsample and r >= 16:
            g = torch.randn(B, cout, r, r, device=dev)
            xl = torch.randn(B, cin, r // 2, r // 2, device=dev)
            res = []
            for mode in ("folded", "materialised"):
                def run():
                    if mode == "folded":
                        return ops.conv2d_wgrad(g, xl, cout, cin, 3, 1, upsample=True)
                    return ops.conv2d_wgrad(g, ops.upsample2x_bilinear(xl), cout, cin, 3, 1)
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    run()
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / 5)
            fl = 2 * 9 * cin * cout * r * r * B
            print(f"   up {cin:4d}->{cout:4d} @{r:3d}^2: folded {res[0] * 1e3:8.1f} us {fl / res[0] / 1e9:6.1f} TF | upsample + wgrad {res[1] * 1e3:8.1f} us {fl / res[1] / 1e9:6.1f} TF")
        cin, r = cout, r * 2
    print(f"total {tot_t:.2f} ms, {tot_f / tot_t / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
