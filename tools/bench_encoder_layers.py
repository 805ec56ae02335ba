#!/usr/bin/env python3
"""Per-layer timing of the grouped ResNet-50 trunk convs (3 encoders x 2 images = 6 groups, batch 8, 256^2 input):
forward (input affine + ReLU, BatchNorm statistics in the epilogue), data gradient and weight gradient of every
distinct conv shape, with how often the shape occurs -- where the generator step's encoder time goes.

    python tools/bench_encoder_layers.py [--batch 8] [--groups 6]
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd")
ops = pkg.ops

# (Cin, Cout, k, stride, Hin, count) of torchvision's ResNet-50 (stride on the 3x3) at a 256^2 image
SHAPES = [
    (3, 64, 7, 2, 256, 1),
    (64, 64, 1, 1, 64, 1), (256, 64, 1, 1, 64, 2), (64, 64, 3, 1, 64, 3), (64, 256, 1, 1, 64, 4),
    (256, 128, 1, 1, 64, 1), (128, 128, 3, 2, 64, 1), (128, 512, 1, 1, 32, 4), (256, 512, 1, 2, 64, 1),
    (512, 128, 1, 1, 32, 3), (128, 128, 3, 1, 32, 3),
    (512, 256, 1, 1, 32, 1), (256, 256, 3, 2, 32, 1), (256, 1024, 1, 1, 16, 6), (512, 1024, 1, 2, 32, 1),
    (1024, 256, 1, 1, 16, 5), (256, 256, 3, 1, 16, 5),
    (1024, 512, 1, 1, 16, 1), (512, 512, 3, 2, 16, 1), (512, 2048, 1, 1, 8, 3), (1024, 2048, 1, 2, 16, 1),
    (2048, 512, 1, 1, 8, 2), (512, 512, 3, 1, 8, 2),
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
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--groups", type=int, default=6)
    args = ap.parse_args()
    dev, B, G = torch.device("cuda:0"), args.batch, args.groups
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    totf = 0.0
    print(f"{'Cin':>5s} {'Cout':>5s} k s {'Hin':>4s} {'n':>2s} | {'fwd us':>8s} {'TF':>6s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}")
    for Cin, Cout, k, s, H, n in SHAPES:
        Ho = ops.conv_out_size(H, k, s)
        first = Cin == 3
        x = torch.randn(B, Cin if first else G * Cin, H, H, device=dev)
        ws = [torch.randn(Cout, Cin, k, k, device=dev) * 0.05 for _ in range(G)]
        cfg = ops.conv2d_pick_config(k, s, B, Cin, Cout, Ho, Ho)
        wp = torch.cat([ops.pack_conv_weight(w, cfg) for w in ws])
        aff = None if first else (torch.rand(G * Cin, device=dev) + 0.5, torch.randn(G * Cin, device=dev) * 0.1)
        stats = torch.zeros(ops.stats_slots(cfg, k, s, B, Cin, Cout, Ho, Ho) * 2 * G * Cout, device=dev, dtype=torch.float64)
        flops = 2.0 * k * k * Cin * Cout * Ho * Ho * B * G
        t_f = timeit(lambda: ops.conv2d_fused(x, wp, Cout, k, s, in_affine=aff, stats=stats, config=cfg, groups=G,
                                              shared_input=first))
        g = torch.randn(B, G * Cout, Ho, Ho, device=dev)
        t_w = timeit(lambda: ops.conv2d_wgrad(g, x, Cout, Cin, k, s, in_affine=aff, groups=G, shared_input=first))
        if first:
            t_d = 0.0
        elif k == 3 and s == 1 and ops.use_wino(B, Cout, Cin, H, H, groups=G):
            # as encoder._GroupedConvBN.conv_bwd runs it: the G trunks as groups of one fp32 Winograd launch
            nw = ops.L.lib().spk_conv2d_packed_bytes_wino(Cout, Cin) // 4
            wtw = torch.empty(G * nw, device=dev)
            ops.pack_conv_weights_wino_into(ws, [wtw[i * nw:(i + 1) * nw] for i in range(G)], transpose_flip=True)
            t_d = timeit(lambda: ops.conv3x3_wino(g, wtw, Cin, groups=G))
        else:
            cfd, tf = ops.dgrad_plan(k, s, B, Cout, Cin, (H, H), (Ho, Ho))
            wt = torch.cat([ops.pack_conv_weight(w, cfd, tf) for w in ws])
            # (a strided 1x1 -- the downsample convs: as encoder._backward runs it since round 3, the gradient stays at the conv's
            # output size and conv1's data gradient adds it at the even pixels in its epilogue; no dilated copy)
            t_d = timeit(lambda: ops.conv2d_dgrad(g, wt, Cin, k, s, (H, H), cfd, groups=G, dilate=not (k == 1 and s == 2)))
        tf = lambda t: flops / (t * 1e-3) / 1e12 if t else 0.0
        print(f"{Cin:5d} {Cout:5d} {k} {s} {H:4d} {n:2d} | {t_f * 1e3:8.1f} {tf(t_f):6.1f} | {t_d * 1e3:8.1f} {tf(t_d):6.1f} | "
              f"{t_w * 1e3:8.1f} {tf(t_w):6.1f}", flush=True)
        tot["fwd"] += n * t_f
        tot["dgrad"] += n * t_d
        tot["wgrad"] += n * t_w
        totf += n * flops
    print("totals (ms): " + ", ".join(f"{k} {v:.2f} ({totf / (v * 1e-3) / 1e12:.1f} TF)" for k, v in tot.items()))


if __name__ == "__main__":
    main()
