#!/usr/bin/env python3
"""Decoder forward (StyleGenerator, 256^2) over batch sizes, exact fp32 and the opt-in bf16x3 path, eager launch plans.

    python tools/bench_batch_sweep.py
"""
import importlib, sys, torch, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd")
dev = torch.device("cuda:0")
torch.manual_seed(0)
g = pkg.StyleGenerator(6144).eval().to(dev)
with torch.no_grad():
    for B in (1, 2, 4, 8, 16, 32):
        x = torch.randn(B, 6144, device=dev)
        for prec in ("f32", "bf16x3"):
            g.synthesis.precision = prec
            for _ in range(3): g(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): g(x)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print(f"B={B:3d} {prec:7s}: {ms:7.3f} ms/step  {B / ms * 1e3:8.1f} frames/s")
