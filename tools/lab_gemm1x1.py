#!/usr/bin/env python3
"""Where the time of the trunk's 1x1 GEMM-form convs goes: the layer timed whole and with parts knocked out (a lab build
of the library: conv1x1_gemm.hip compiled with -DSPK_GEMM_LAB, selected with SPK_LAB_LIB), next to the plain streaming
floor of the same bytes (a device copy of the input's size plus a fill of the output's size).

    SPK_LAB_LIB=tools/_bin/libspk_hip_lab.so python tools/lab_gemm1x1.py
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

SHAPES = [(64, 256, 64), (256, 64, 64), (256, 128, 64), (128, 512, 32), (512, 128, 32), (512, 256, 32), (256, 1024, 16), (1024, 256, 16)]


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


def main():
    dev, B, G = torch.device("cuda:0"), 8, 6
    print(f"{'Cin':>5s} {'Cout':>5s} {'H':>3s} cfg | {'whole':>7s} {'noMFMA':>7s} {'noLoad':>7s} {'noStore':>7s} {'ldonly':>7s} {'mfma':>7s} {'m-stats':>7s} {'m-epi':>7s} {'nostats':>7s} | {'copy+fill':>9s} {'MB in/out':>12s}")
    for Cin, Cout, H in SHAPES:
        x = torch.randn(B, G * Cin, H, H, device=dev)
        ws = [torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05 for _ in range(G)]
        picked = ops.conv2d_pick_config(1, 1, B, Cin, Cout, H, H)
        cfg = 12
        wp = torch.cat([ops.pack_conv_weight(w, cfg) for w in ws])
        sc = torch.rand(G * Cin, device=dev) + 0.5
        sh = torch.randn(G * Cin, device=dev) * 0.1
        y = torch.empty(B, G * Cout, H, H, device=dev)

        stats = torch.zeros(ops.stats_slots(cfg, 1, 1, B, Cin, Cout, H, H) * 2 * G * Cout, device=dev, dtype=torch.float64)

        def fwd():
            ops.conv2d_fused(x, wp, Cout, 1, 1, in_affine=(sc, sh), stats=stats, config=cfg, groups=G, out=y)

        res = []
        for mask in (0, 1, 2, 4, 5, 6, 6 | 16, 6 | 8, 16):
            os.environ["SPK_GEMM_LAB"] = str(mask)
            res.append(timeit(fwd))
        os.environ["SPK_GEMM_LAB"] = "0"
        x2 = torch.empty_like(x)

        def floor():
            x2.copy_(x)
            y.fill_(1.0)

        def rd():
            x2.copy_(x)

        fl = timeit(floor)
        mb_in, mb_out = x.numel() * 4 / 1e6, y.numel() * 4 / 1e6
        print(f"{Cin:5d} {Cout:5d} {H:3d} {picked:3d} | {res[0]:7.1f} {res[1]:7.1f} {res[2]:7.1f} {res[3]:7.1f} {res[4]:7.1f} {res[5]:7.1f} {res[6]:7.1f} {res[7]:7.1f} {res[8]:7.1f} | {fl:9.1f} {mb_in:5.0f}/{mb_out:5.0f}", flush=True)


if __name__ == "__main__":
    main()
