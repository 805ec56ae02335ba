#!/usr/bin/env python3
"""Which ATen ops (not our HIP kernels) a generator / discriminator step still runs: torch.profiler table by op and shape.

    python tools/profile_gstep_ops.py [--d-step]
"""
import argparse
import importlib
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d-step", action="store_true")
    args = ap.parse_args()
    import bench
    pkg = importlib.import_module("speak-hack_amd")
    dev = torch.device("cuda:0")
    # reuse bench.irfd_steps' step by running it under the profiler for 2 steps after its warm-up
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        bench.irfd_steps(pkg, dev, "d" if args.d_step else "g", 8, steps=2, warmup=2)
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
    rows.sort(key=lambda e: -e.device_time_total)
    print(f"{'op':40s} {'calls':>6s} {'gpu ms':>9s}  shapes")
    for e in rows[:45]:
        print(f"{e.key[:40]:40s} {e.count:6d} {e.device_time_total / 1e3:9.3f}  {str(e.input_shapes)[:110]}")


if __name__ == "__main__":
    main()
