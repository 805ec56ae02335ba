#!/usr/bin/env python3
"""LAB: what a fusion could gain at most.  Times the generator step (B=8) and the discriminator step (B=8) of bench.py with one
memory-bound helper pass at a time KNOCKED OUT (its launch skipped, its output left uninitialised -- the numbers the step computes
are then wrong; only the time is read).  The difference to the untouched step is the upper bound of what folding that pass into a
neighbouring MFMA kernel can win, BEFORE the cost of the extra epilogue / staging work it adds there.

    python tools/lab_knockout.py [--steps 6]
"""
import argparse
import importlib
import importlib.util
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pkg = importlib.import_module("speak-hack_amd")
    ops, L = pkg.ops, pkg._lib
    dev = torch.device("cuda:0")
    real = {k: getattr(ops, k) for k in ("bn_backward", "epilogue_bwd", "upsample2x_bilinear_bwd", "conv2d_wgrad", "bn_add_relu",
                                         "maxpool3x3s2_bwd")}

    def bn_no_reduce(g, r, affine, mean, invstd, mask_mode, mask_src=None, g_scale=1.0, g_per_plane=False, want_dz=False,
                     batch_stats=True):
        B, Cc, H, W = r.shape
        HW = H * W
        sums = torch.empty((B, 2, Cc), device=r.device, dtype=torch.float32)
        args_ = (L.dptr(g, "g"), L.dptr(r, "r"), L.dptr(mask_src, "mask_src"), int(mask_mode), L.dptr(affine[0], "scale"),
                 L.dptr(affine[1], "shift"), L.dptr(mean, "mean"), L.dptr(invstd, "invstd"))
        csum = torch.empty((2, Cc), device=r.device, dtype=torch.float32)
        dr = torch.empty_like(r)
        dz = torch.empty_like(r) if want_dz else None
        L.check(L.lib().spk_bn_bwd_apply_sums(*args_, L.dptr(sums), L.dptr(csum), 1 if batch_stats else 0, B * HW, float(g_scale),
                                              1 if g_per_plane else 0, L.dptr(dr), L.dptr(dz), B, Cc, HW, L.stream_ptr()), "apply")
        out = (dr, csum[1], csum[0])
        return out + (dz,) if want_dz else out

    def bn_nothing(g, r, affine, mean, invstd, mask_mode, mask_src=None, g_scale=1.0, g_per_plane=False, want_dz=False,
                   batch_stats=True):
        Cc = r.shape[1]
        csum = torch.empty((2, Cc), device=r.device, dtype=torch.float32)
        dr = g if (not g_per_plane and g.shape == r.shape) else torch.empty_like(r)
        out = (dr, csum[1], csum[0])
        return out + (dr,) if want_dz else out

    def epi_nothing(dy, a=None, noise=None, style=None, slope=1.0, inplace=False):
        B, Cc = dy.shape[:2]
        return dy, torch.empty((B, 4, Cc), device=dy.device, dtype=torch.float32)

    def ups_nothing(dy):
        B, Cc, H2, W2 = dy.shape
        return torch.empty((B, Cc, H2 // 2, W2 // 2), device=dy.device, dtype=torch.float32)

    def wgrad_nothing(g, x, Cout, Cin, k=3, stride=1, *, out=None, groups=1, fold=1, **kw):
        if out is not None:
            return out
        return torch.empty((int(groups) // int(fold) * Cout, Cin, k, k), device=g.device, dtype=torch.float32)

    cases = [("base", {}), ("bn_bwd without the reduce pass", {"bn_backward": bn_no_reduce}),
             ("bn_bwd without both passes", {"bn_backward": bn_nothing}),
             ("no epilogue_bwd", {"epilogue_bwd": epi_nothing}), ("no upsample2x_bwd", {"upsample2x_bilinear_bwd": ups_nothing}),
             ("no weight gradients at all", {"conv2d_wgrad": wgrad_nothing}), ("base again", {})]
    res = {}
    for which, B in (("g", 8), ("d", 8)):
        for name, patch in cases:
            if args.only and args.only not in name:
                continue
            if which == "d" and name.startswith(("bn_bwd", "no upsample")):
                continue
            for k, v in real.items():
                setattr(ops, k, patch.get(k, v))
            try:
                ms = bench.irfd_steps(pkg, dev, which, B, steps=args.steps, warmup=2)
            finally:
                for k, v in real.items():
                    setattr(ops, k, v)
            res[f"{which}-step B={B}: {name}"] = round(ms, 2)
            print(f"{which}-step B={B}: {name:40s} {ms:8.2f} ms", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
