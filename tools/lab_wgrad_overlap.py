#!/usr/bin/env python3
"""Does a weight gradient on a SECOND HIP stream hide behind the data-gradient chain?  [dgrad(i); wgrad(i)] x n on one stream
against dgrad(i) on the main stream and wgrad(i) on a side stream (event after the producer, one join at the end).

    python tools/lab_wgrad_overlap.py
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd")
ops = pkg.ops

# (Cin, Cout, k, H, B, groups, affine)
CASES = [(64, 64, 3, 256, 8, 1, 0), (128, 128, 3, 128, 8, 1, 0), (256, 256, 3, 64, 8, 1, 0), (512, 512, 3, 32, 8, 1, 0), (512, 512, 3, 16, 8, 1, 0),
         (256, 256, 3, 64, 16, 1, 0), (512, 512, 3, 32, 16, 1, 0), (128, 128, 3, 128, 16, 1, 0),
         (256, 1024, 1, 16, 8, 6, 1), (1024, 256, 1, 16, 8, 6, 1), (256, 256, 3, 16, 8, 6, 1), (64, 64, 3, 64, 8, 6, 1)]


def main():
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream(dev)
    n = 20
    print(f"{'case':>34s} | dgrad+wgrad serial us | two streams us | gain")
    for Cin, Cout, k, H, B, G, aff in CASES:
        x = torch.randn(B, G * Cin, H, H, device=dev)
        g = torch.randn(B, G * Cout, H, H, device=dev)
        ws = [torch.randn(Cout, Cin, k, k, device=dev) * 0.05 for _ in range(G)]
        cfg, _ = ops.dgrad_plan(k, 1, B, Cout, Cin, (H, H), (H, H))
        wp = ops.pack_conv_weights_list(ws, cfg, transpose_flip=True)
        a = (torch.rand(G * Cin, device=dev) + 0.5, torch.randn(G * Cin, device=dev) * 0.1) if aff else None
        dx = torch.empty_like(x)
        dw = torch.empty(G * Cout, Cin, k, k, device=dev)

        def dgrad():
            ops.conv2d_fused(g, wp, Cin, k, 1, config=cfg, groups=G, out=dx)

        def wgrad():
            ops.conv2d_wgrad(g, x, Cout, Cin, k, 1, in_affine=a, groups=G, out=dw)

        def serial():
            for _ in range(n):
                dgrad()
                wgrad()

        def overlapped():
            main_s = torch.cuda.current_stream(dev)
            for _ in range(n):
                ev = torch.cuda.Event()
                ev.record(main_s)
                dgrad()
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    wgrad()
            main_s.wait_stream(side)

        def joined():                                # safe form: both kernels of a layer start together, main waits for both
            main_s = torch.cuda.current_stream(dev)
            for _ in range(n):
                side.wait_stream(main_s)
                with torch.cuda.stream(side):
                    wgrad()
                dgrad()
                main_s.wait_stream(side)

        res = []
        for fn in (serial, overlapped, joined):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / (3 * n) * 1e3)
        print(f"{str((Cin, Cout, k, H, B, G)):>34s} | {res[0]:21.1f} | {res[1]:14.1f} | {res[0] / res[1]:.3f}x | joined per layer {res[2]:8.1f} {res[0] / res[2]:.3f}x", flush=True)


if __name__ == "__main__":
    main()
