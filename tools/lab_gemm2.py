#!/usr/bin/env python3
"""Where the time of the three-per-CU 1x1 GEMM form (conv1x1_gemm2.hip, configs 14 / 15) goes: the trunk's layers timed whole
and with parts knocked out at COMPILE time (a lab build of the library: conv1x1_gemm2.hip with -DSPK_G2_LAB, which adds the
knock-out instantiations; SPK_G2_LAB=<mask> picks one per launch), next to the streaming floor of the same bytes.

    hipcc ... -DSPK_G2_LAB -c speak-hack_amd/csrc/conv1x1_gemm2.hip -o tools/_bin/conv1x1_gemm2_lab.o   (+ link: see DESIGN)
    SPK_LAB_LIB=tools/_bin/libspk_hip_g2lab.so python tools/lab_gemm2.py

mask bits: 1 no MFMAs, 2 no epilogue, 4 no x DMA, 8 no weight DMA, 16 no fragment reads, 64 no barrier, 128 time stamps.
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd")
if os.environ.get("SPK_LAB_LIB"):
    pkg._lib.LIB_PATH = os.path.abspath(os.environ["SPK_LAB_LIB"])
ops = pkg.ops

SHAPES = [(64, 256, 64), (256, 64, 64), (256, 128, 64), (128, 512, 32), (512, 128, 32), (512, 256, 32), (256, 1024, 16),
          (1024, 256, 16), (512, 2048, 8)]
MASKS = [(0, "whole"), (2, "noEpi"), (1, "noMFMA"), (3, "mem+lds"), (4, "noX"), (8, "noW"), (12, "noGlob"), (14, "kloop"),
         (30, "k-noFrag"), (94, "mfma"), (13, "epi"), (64, "noBar"), (16, "noFrag")]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def stagger_scan():
    """LAB_STAGGER=0,50,100,...: the whole kernel per first-round stagger (units of 64 cycles per CU slot)."""
    dev, B, G = torch.device("cuda:0"), 8, 6
    vals = [int(v) for v in os.environ["LAB_STAGGER"].split(",")]
    cfgs = [int(c) for c in os.environ.get("LAB_CFGS", "14").split(",")]
    print(f"{'Cin':>5s} {'Cout':>5s} {'H':>3s} cfg | " + " ".join(f"{v:>7d}" for v in vals) + " | ideal(140TF)")
    for Cin, Cout, H in SHAPES:
        for cfg in cfgs:
            x = torch.randn(B, G * Cin, H, H, device=dev)
            ws = [torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05 for _ in range(G)]
            wp = ops.pack_conv_weights_list(ws, cfg)
            sc = torch.rand(G * Cin, device=dev) + 0.5
            sh = torch.randn(G * Cin, device=dev) * 0.1
            y = torch.empty(B, G * Cout, H, H, device=dev)
            stats = torch.zeros(ops.stats_slots(cfg, 1, 1, B, Cin, Cout, H, H) * 2 * G * Cout, device=dev, dtype=torch.float64)

            def fwd():
                ops.conv2d_fused(x, wp, Cout, 1, 1, in_affine=(sc, sh), stats=stats, config=cfg, groups=G, out=y)

            res = []
            for v in vals:
                os.environ["SPK_G2_STAGGER"] = str(v)
                res.append(timeit(fwd))
            ideal = 2.0 * Cin * Cout * H * H * B * G / 140e12 * 1e6
            print(f"{Cin:5d} {Cout:5d} {H:3d} {cfg:3d} | " + " ".join(f"{r:7.1f}" for r in res) + f" | {ideal:8.1f}", flush=True)


def main():
    if os.environ.get("LAB_STAGGER"):
        return stagger_scan()
    dev, B, G = torch.device("cuda:0"), 8, 6
    cfgs = [int(c) for c in os.environ.get("LAB_CFGS", "14").split(",")]
    print(f"{'Cin':>5s} {'Cout':>5s} {'H':>3s} cfg | " + " ".join(f"{n:>8s}" for _, n in MASKS) + " | copy+fill  ideal(140TF)")
    for Cin, Cout, H in SHAPES:
        for cfg in cfgs:
            x = torch.randn(B, G * Cin, H, H, device=dev)
            ws = [torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05 for _ in range(G)]
            wp = ops.pack_conv_weights_list(ws, cfg)
            sc = torch.rand(G * Cin, device=dev) + 0.5
            sh = torch.randn(G * Cin, device=dev) * 0.1
            y = torch.empty(B, G * Cout, H, H, device=dev)
            stats = torch.zeros(ops.stats_slots(cfg, 1, 1, B, Cin, Cout, H, H) * 2 * G * Cout, device=dev, dtype=torch.float64)

            def fwd():
                ops.conv2d_fused(x, wp, Cout, 1, 1, in_affine=(sc, sh), stats=stats, config=cfg, groups=G, out=y)

            res = []
            for mask, _ in MASKS:
                os.environ["SPK_G2_LAB"] = str(mask)
                res.append(timeit(fwd))
            os.environ["SPK_G2_LAB"] = "0"
            x2 = torch.empty_like(x)

            def floor():
                x2.copy_(x)
                y.fill_(1.0)

            fl = timeit(floor)
            ideal = 2.0 * Cin * Cout * H * H * B * G / 140e12 * 1e6
            print(f"{Cin:5d} {Cout:5d} {H:3d} {cfg:3d} | " + " ".join(f"{r:8.1f}" for r in res) + f" | {fl:9.1f} {ideal:8.1f}", flush=True)


if __name__ == "__main__":
    main()
