#!/usr/bin/env python3
"""Per-layer timing of the decoder's 3x3 convs (B = 8) on the opt-in bf16x3 kernel against the exact f32 kernel.

    python tools/bench_bf16x3.py [--batch 8] [--res 256]
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd")
ops = pkg.ops


def timed(fn, reps=10):
    for _ in range(3):
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
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--plain-epilogue", action="store_true", help="no bias / noise / lrelu / style")
    args = ap.parse_args()
    dev, B = torch.device("cuda:0"), args.batch
    r, cin, tot = 8, 512, [0.0, 0.0, 0.0]
    while r <= args.res:
        cout = min(int(8192 / (2.0 ** (r.bit_length() - 2))), 512)
        for ci, co, up in ((cin, cout, True), (cout, cout, False)):
            hin = r // 2 if up else r
            x = torch.randn(B, ci, hin, hin, device=dev)
            w = torch.randn(co, ci, 3, 3, device=dev) * (9 * ci) ** -0.5
            bias, nw = torch.randn(co, device=dev), torch.randn(co, device=dev)
            nz, st = torch.randn(B, 1, r, r, device=dev), torch.randn(B, 2 * co, device=dev)
            kw = {} if args.plain_epilogue else dict(bias=bias, noise_w=nw, noise=nz, style=st, lrelu_slope=0.2)
            cfg = ops.conv2d_pick_config(3, 1, B, ci, co, r, r)
            wp = ops.pack_conv_weight(w, cfg)
            out = torch.empty(B, co, r, r, device=dev)
            t32 = timed(lambda: ops.conv2d_fused(x, wp, co, 3, 1, upsample=up, config=cfg, out=out, **kw))
            fl = 2 * 9 * ci * co * r * r * B
            line = f"{ci:4d}->{co:4d} @{r:3d}^2 {'up' if up else '  '}: f32 {t32 * 1e3:7.1f} us {fl / t32 / 1e9:6.1f} TF"
            tot[0] += t32
            tot[2] += fl
            if ops.bf16x3_supported(B, ci, co, r, r):
                wb = ops.pack_conv_weight_bf16x3(w)
                tb = timed(lambda: ops.conv3x3_bf16x3(x, wb, co, upsample=up, out=out, **kw))
                line += f" | bf16x3 {tb * 1e3:7.1f} us {fl / tb / 1e9:6.1f} TF (x{t32 / tb:.2f})"
                tot[1] += tb if B * r * r >= 8192 else t32
            print(line)
        cin, r = cout, r * 2
    print(f"total: f32 {tot[0]:.2f} ms ({tot[2] / tot[0] / 1e9:.1f} TF), mixed (bf16x3 from 8192 pixels) {tot[1]:.2f} ms ({tot[2] / tot[1] / 1e9:.1f} TF)")


if __name__ == "__main__":
    main()
