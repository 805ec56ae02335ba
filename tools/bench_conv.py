#!/usr/bin/env python3
"""Per-layer A/B of the conv tile configs on the GPU box (interleaved rounds in one process).

    python tools/bench_conv.py [--batch 8] [--res 256]

Prints, for every 3x3 conv of the decoder, the time and TFLOP/s of each (config, ksplit) candidate.
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd")
ops = pkg.ops


def layers(res):
    out, r, cin = [], 8, 512
    while r <= res:
        cout = min(int(8192 / (2.0 ** (r.bit_length() - 2))), 512)
        out += [(cin, cout, r, True), (cout, cout, r, False)]
        cin, r = cout, r * 2
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--dgrad", action="store_true", help="the data-gradient shapes of the same layers (channels swapped, plain)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    ncfg = pkg._lib.lib().spk_conv2d_num_configs()
    shapes = [(co, ci, r, False) for ci, co, r, _ in layers(args.res) if ci != co] if args.dgrad else layers(args.res)
    for cin, cout, r, ups in shapes:
        hs = r // 2 if ups else r
        x = torch.randn(B, cin, hs, hs, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
        bias = torch.randn(cout, device=dev)
        nw = torch.randn(cout, device=dev)
        nz = torch.randn(B, 1, r, r, device=dev)
        st = torch.randn(B, 2 * cout, device=dev)
        flops = 2 * 9 * cin * cout * r * r * B
        cands = []
        for cfg in range(ncfg):
            if not ops.conv3x3_config_fits(cfg, B, cin, cout, r, r):
                continue
            co_t, ci_t, px_t = ops.conv3x3_config_info(cfg)
            if co_t > 2 * cout:
                continue
            wp = ops.pack_conv3x3_weight(w, cfg)
            for ks in (1, 0, 2, 4, 8, 16):
                cands.append((cfg, ks, wp))
        times = {i: [] for i in range(len(cands))}
        for rd in range(args.rounds + 1):
            for i, (cfg, ks, wp) in enumerate(cands):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    ops.conv3x3_fused(x, wp, cout, bias=bias, noise_w=nw, noise=nz, style=st, upsample=ups,
                                      lrelu_slope=0.2, config=cfg, ksplit=ks)
                e1.record()
                torch.cuda.synchronize()
                if rd > 0:
                    times[i].append(e0.elapsed_time(e1) / 3)
        auto = ops.conv3x3_pick_config(B, cin, cout, r, r)
        print(f"--- {cin:3d}->{cout:3d} @{r:3d}^2 ups={int(ups)} B={B}  {flops / 1e9:7.2f} GFLOP   (auto config {auto})")
        res = sorted(((sorted(times[i])[len(times[i]) // 2], cands[i][0], cands[i][1]) for i in times))
        for t, cfg, ks in res[:6]:
            print(f"    cfg {cfg} ksplit {ks:2d}: {t * 1e3:8.1f} us  {flops / t / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
