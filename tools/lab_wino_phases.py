#!/usr/bin/env python3
"""LAB: cycle stamps of one workgroup's SECOND region in the Winograd kernel (a -DLAB build of csrc/conv3x3_wino_f32.hip that writes
the stamps over the first floats of the output): region start -> K loop start -> K loop end -> half 0 transformed -> half 0 stored ->
half 1 transformed -> half 1 stored.   SPK_LAB_LIB=tools/_bin/libspk_hip_winolab.so python tools/lab_wino_phases.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd")
pkg._lib.LIB_PATH = os.path.abspath(os.environ["SPK_LAB_LIB"])
ops = pkg.ops
dev = torch.device("cuda:0")
names = ["region start", "K loop start", "K loop end", "table published", "rows stored"]
for Cin, Cout, R in ((64, 64, 256), (128, 64, 256), (128, 128, 128), (512, 512, 64)):
    x = torch.randn(8, Cin, R, R, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    bias, nw = torch.randn(Cout, device=dev), torch.randn(Cout, device=dev)
    noise, style = torch.randn(8, 1, R, R, device=dev), torch.randn(8, 2 * Cout, device=dev)
    ww = ops.pack_conv_weight_wino(w)
    for _ in range(3):
        y = ops.conv3x3_wino(x, ww, Cout, bias=bias, noise_w=nw, noise=noise, style=style, lrelu_slope=0.2)
    torch.cuda.synchronize()
    t = y.flatten()[:5].tolist()
    print(f"{Cin}->{Cout} @{R}^2: " + "; ".join(f"{n} {int(v)}" for n, v in zip(names, t)) + f"   (chunks {Cin // 8}: MFMA floor {Cin // 8 * 4096} cycles)")
