#!/usr/bin/env python3
"""Which Python lines issue which ATen ops during one training step (a TorchDispatchMode count by call site): the small
launches of a step that are not our HIP kernels.

    python tools/op_callsites.py [--d-step] [--batch 8]
"""
import argparse
import collections
import importlib
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        site = "?"
        for fr in reversed(traceback.extract_stack(limit=24)):
            fn = fr.filename
            if fn.startswith(ROOT) and "op_callsites" not in fn:
                site = f"{os.path.relpath(fn, ROOT)}:{fr.lineno}"
                break
        self.counts[(name, site)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d-step", action="store_true")
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    import bench
    pkg = importlib.import_module("speak-hack_amd")
    dev = torch.device("cuda:0")
    c = Counter()
    bench.irfd_steps(pkg, dev, "d" if args.d_step else "g", args.batch, steps=1, warmup=2, observe=c)
    skip = ("empty", "view", "detach", "as_strided", "_unsafe_view", "t.default", "transpose", "expand", "slice", "select", "alias",
            "unsqueeze", "squeeze", "permute", "reshape", "_local_scalar", "split", "unbind", "lift_fresh", "is_same_size")
    rows = [(k, v) for k, v in c.counts.items() if not any(s in k[0] for s in skip)]
    rows.sort(key=lambda kv: -kv[1])
    print(f"{sum(v for _, v in rows)} launching ATen calls in one step")
    for (name, site), v in rows[:60]:
        print(f"{v:5d}  {name:34s} {site}")


if __name__ == "__main__":
    main()
