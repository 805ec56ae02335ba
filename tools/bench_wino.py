#!/usr/bin/env python3
"""Per-layer time of the fp32 Winograd conv (SPK_CONV_WINOGRAD) beside the direct f32 MFMA kernel on the decoder's 3x3 layers
(and the discriminator's), batch 8.  Plain input for both (the x2 layers read a materialised upsampled tensor here)."""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ev_ms(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    pkg = importlib.import_module("speak-hack_amd")
    if os.environ.get("SPK_LAB_LIB"):          # A/B against another build of the library
        pkg._lib.LIB_PATH = os.path.abspath(os.environ["SPK_LAB_LIB"])
    ops = pkg.ops
    dev = torch.device("cuda:0")
    B = args.batch
    layers = [(512, 512, 32), (512, 512, 64), (512, 256, 64), (256, 256, 64), (256, 128, 128), (128, 128, 128), (128, 64, 256), (64, 64, 256),
              (64, 64, 128), (128, 128, 64)]
    tot_d = tot_w = 0.0
    print(f"{'layer':>22s} {'direct us':>10s} {'TF/s':>7s} {'wino us':>9s} {'alg TF/s':>9s} {'exec frac':>9s} {'ratio':>6s}")
    for Cin, Cout, R in layers:
        if not ops.wino_supported(B, Cin, Cout, R, R):
            continue
        x = torch.randn(B, Cin, R, R, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        bias, nw = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev)
        noise, style = torch.randn(B, 1, R, R, device=dev), torch.randn(B, 2 * Cout, device=dev)
        out = torch.empty(B, Cout, R, R, device=dev)
        cfg = ops.conv2d_pick_config(3, 1, B, Cin, Cout, R, R)
        wd, ww = ops.pack_conv_weight(w, cfg), ops.pack_conv_weight_wino(w)
        kw = dict(bias=bias, noise_w=nw, noise=noise, style=style, lrelu_slope=0.2, out=out)
        td = ev_ms(lambda: ops.conv2d_fused(x, wd, Cout, 3, 1, config=cfg, **kw), args.reps) * 1e3
        tw = ev_ms(lambda: ops.conv3x3_wino(x, ww, Cout, **kw), args.reps) * 1e3
        fl = 2 * 9 * Cin * Cout * R * R * B
        tot_d += td
        tot_w += tw
        print(f"{Cin:4d}->{Cout:4d} @{R:3d}^2 B={B} {td:10.1f} {fl / td / 1e6:7.1f} {tw:9.1f} {fl / tw / 1e6:9.1f} {fl * 16 / 36 / tw / 1e6 / 157.3:9.3f} {td / tw:6.2f}")
    print(f"{'total':>22s} {tot_d:10.1f} {'':7s} {tot_w:9.1f}   ratio {tot_d / tot_w:.2f}")


if __name__ == "__main__":
    main()
