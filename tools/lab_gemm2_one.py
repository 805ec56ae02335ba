#!/usr/bin/env python3
"""One layer of tools/lab_gemm2.py, N launches per knock-out mask -- the thing to put under rocprofv3:
    SPK_LAB_LIB=tools/_bin/libspk_hip_g2lab.so python tools/lab_gemm2_one.py Cin Cout H cfg mask[,mask...] [launches]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd")
if os.environ.get("SPK_LAB_LIB"):
    pkg._lib.LIB_PATH = os.path.abspath(os.environ["SPK_LAB_LIB"])
ops = pkg.ops
Cin, Cout, H, cfg = (int(v) for v in sys.argv[1:5])
masks = [int(m) for m in sys.argv[5].split(",")]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dev, B, G = torch.device("cuda:0"), 8, 6
x = torch.randn(B, G * Cin, H, H, device=dev)
ws = [torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05 for _ in range(G)]
wp = ops.pack_conv_weights_list(ws, cfg)
sc = torch.rand(G * Cin, device=dev) + 0.5
sh = torch.randn(G * Cin, device=dev) * 0.1
y = torch.empty(B, G * Cout, H, H, device=dev)
stats = torch.zeros(ops.stats_slots(cfg, 1, 1, B, Cin, Cout, H, H) * 2 * G * Cout, device=dev, dtype=torch.float64)
for m in masks:
    os.environ["SPK_G2_LAB"] = str(m)
    for _ in range(n):
        ops.conv2d_fused(x, wp, Cout, 1, 1, in_affine=(sc, sh), stats=stats, config=cfg, groups=G, out=y)
    torch.cuda.synchronize()
print("done")
