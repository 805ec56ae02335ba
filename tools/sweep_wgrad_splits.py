#!/usr/bin/env python3
"""Weight-gradient time against the number of pixel splits (slabs), per layer shape: the data behind the split rule of
`wgrad_mfma_f32.hip` (`pick_splits`).  A split count trades whole rounds of one workgroup per CU (256 CUs) against
slab traffic (every split writes, and the reduce reads, one [Cout, Cin, taps] slab).

    python tools/sweep_wgrad_splits.py [--set trunk|decoder|disc] [--batch 8]
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd")
ops = pkg.ops

# (Cin, Cout, k, stride, Hin, groups, affine)
TRUNK = [
    (64, 64, 3, 1, 64, 6, 1), (128, 128, 3, 2, 64, 6, 1), (128, 128, 3, 1, 32, 6, 1), (256, 256, 3, 2, 32, 6, 1),
    (256, 256, 3, 1, 16, 6, 1), (512, 512, 3, 2, 16, 6, 1), (512, 512, 3, 1, 8, 6, 1),
    (256, 64, 1, 1, 64, 6, 1), (64, 256, 1, 1, 64, 6, 1), (128, 512, 1, 1, 32, 6, 1), (512, 128, 1, 1, 32, 6, 1),
    (256, 1024, 1, 1, 16, 6, 1), (1024, 256, 1, 1, 16, 6, 1), (512, 2048, 1, 1, 8, 6, 1), (2048, 512, 1, 1, 8, 6, 1),
    (256, 512, 1, 2, 64, 6, 1), (512, 1024, 1, 2, 32, 6, 1), (1024, 2048, 1, 2, 16, 6, 1),
]
DECODER = [  # StyleGenerator, 256^2: (Cin, Cout, 3, 1, H of the conv's output plane)
    (512, 512, 3, 1, 8, 1, 0), (512, 512, 3, 1, 16, 1, 0), (512, 512, 3, 1, 32, 1, 0), (512, 256, 3, 1, 64, 1, 0),
    (256, 256, 3, 1, 64, 1, 0), (256, 128, 3, 1, 128, 1, 0), (128, 128, 3, 1, 128, 1, 0), (128, 64, 3, 1, 256, 1, 0),
    (64, 64, 3, 1, 256, 1, 0),
]
DISC = [  # the discriminator's residual blocks at a 256^2 frame (conv1 3x3 s1, conv2 3x3 s2, skip 1x1 s2)
    (64, 64, 3, 1, 256, 1, 0), (64, 128, 3, 2, 256, 1, 0), (128, 128, 3, 1, 128, 1, 0), (128, 256, 3, 2, 128, 1, 0),
    (256, 256, 3, 1, 64, 1, 0), (256, 512, 3, 2, 64, 1, 0), (512, 512, 3, 1, 32, 1, 0), (512, 512, 3, 2, 32, 1, 0),
    (512, 512, 3, 1, 16, 1, 0), (512, 512, 3, 2, 16, 1, 0),
]


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="trunk")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--splits", default="0,1,2,3,4,6,8,12,16,24,32,48,64")
    args = ap.parse_args()
    dev, B = torch.device("cuda:0"), args.batch
    cand = [int(s) for s in args.splits.split(",")]
    shapes = {"trunk": TRUNK, "decoder": DECODER, "disc": DISC}[args.set]
    print(f"{'Cin':>5s} {'Cout':>5s} k s {'Hin':>4s} {'G':>2s} | " + " ".join(f"{('auto' if s == 0 else s):>7}" for s in cand) + " |  best")
    for Cin, Cout, k, s, Hin, G, aff in shapes:
        H = Hin // s
        x = torch.randn(B, G * Cin, Hin, Hin, device=dev)
        g = torch.randn(B, G * Cout, H, H, device=dev)
        a = (torch.rand(G * Cin, device=dev) + 0.5, torch.randn(G * Cin, device=dev) * 0.1) if aff else None
        res = []
        for sp in cand:
            try:
                res.append(timeit(lambda: ops.conv2d_wgrad(g, x, Cout, Cin, k, s, in_affine=a, groups=G, splits=sp)))
            except Exception:                                    # more splits than tiles, workspace limits
                res.append(float("nan"))
        explicit = [(r, c) for r, c in zip(res, cand) if r == r and c > 0]
        tail = f" | {min(explicit)[1]:3d} ({res[0] / min(explicit)[0]:.2f}x)" if explicit else ""
        print(f"{Cin:5d} {Cout:5d} {k} {s} {Hin:4d} {G:2d} | " + " ".join(f"{r:7.1f}" for r in res) + tail, flush=True)


if __name__ == "__main__":
    main()
