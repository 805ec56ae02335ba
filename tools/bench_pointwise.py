#!/usr/bin/env python3
"""The memory-bound helpers of a generator step at their largest shapes, beside the time their bytes take at 4.5 TB/s:
max-pool forward / adjoint (stem), bilinear x2 adjoint (256^2 decoder layer), toRGB backward.

    python tools/bench_pointwise.py
"""
import importlib, sys, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("speak-hack_amd"); ops = pkg.ops
dev = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(8, 384, 128, 128, device=dev); s = torch.rand(384, device=dev) + .5; o = torch.randn(384, device=dev) * .1
g = torch.randn(8, 384, 64, 64, device=dev)
print("maxpool fwd us", t(lambda: ops.maxpool3x3s2(x, s, o)), "ideal(4.5TB/s)", (x.numel() + g.numel()) * 4 / 4.5e12 * 1e6)
print("maxpool bwd us", t(lambda: ops.maxpool3x3s2_bwd(x, g, s, o)), "ideal", (2 * x.numel() + g.numel()) * 4 / 4.5e12 * 1e6)
dy = torch.randn(16, 128, 256, 256, device=dev)
print("upsample2x bwd us", t(lambda: ops.upsample2x_bilinear_bwd(dy)), "ideal", dy.numel() * 1.25 * 4 / 4.5e12 * 1e6)
x2 = torch.randn(16, 64, 256, 256, device=dev); w = torch.randn(3, 64, 1, 1, device=dev); dy2 = torch.randn(16, 3, 256, 256, device=dev)
print("toRGB bwd (dx+dw) us", t(lambda: ops.conv1x1_small_bwd(x2, w, dy2, need_dx=True)), "ideal", (2 * x2.numel() + 2 * dy2.numel()) * 4 / 4.5e12 * 1e6)
print("toRGB bwd (dw) us", t(lambda: ops.conv1x1_small_bwd(x2, w, dy2, need_dx=False)), "ideal", (x2.numel() + dy2.numel()) * 4 / 4.5e12 * 1e6)
