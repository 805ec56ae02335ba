#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/<dir>) into a small text summary for profiles/.

    python tools/summarize_profile.py gpurun_out/prof_x [more dirs...] > profiles/rNN_name.txt

Handles: *_kernel_stats.csv (per-kernel totals), *_kernel_trace.csv (per-dispatch durations; the
last step's conv launches are listed one by one), *_counter_collection.csv (PMC sums per kernel).
"""
from __future__ import annotations

import csv
import glob
import os
import sys
from collections import defaultdict


def short(name, n=88):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:n]


def kernel_stats(path, out):
    rows = list(csv.DictReader(open(path)))
    out.append(f"## kernel stats ({os.path.basename(path)})")
    out.append(f"{'kernel':88s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for r in rows[:18]:
        out.append(f"{short(r['Name']):88s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:10.3f} "
                   f"{float(r['AverageNs']) / 1e3:10.2f} {float(r['MinNs']) / 1e3:9.2f} {float(r['MaxNs']) / 1e3:9.2f} "
                   f"{float(r['Percentage']):6.2f}")


def kernel_trace(path, out, n_conv=12):
    rows = list(csv.DictReader(open(path)))
    convs = [r for r in rows if ("conv_kernel<" in r["Kernel_Name"] or "conv3x3" in r["Kernel_Name"])
             and "pack" not in r["Kernel_Name"]]
    if not convs:
        return
    out.append(f"## last {n_conv} conv dispatches ({os.path.basename(path)})")
    out.append(f"{'kernel':70s} {'grid(thr)':>12s} {'wg':>5s} {'lds':>7s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'us':>9s}")
    for r in convs[-n_conv:]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        out.append(f"{short(r['Kernel_Name'], 70):70s} {r['Grid_Size_X'] + 'x' + r['Grid_Size_Y']:>12s} "
                   f"{r['Workgroup_Size_X']:>5s} {r['LDS_Block_Size']:>7s} {r['VGPR_Count']:>5s} "
                   f"{r.get('Accum_VGPR_Count', '-'):>5s} {r['SGPR_Count']:>5s} {d:9.1f}")


def counters(path, out):
    rows = list(csv.DictReader(open(path)))
    if not rows:
        return
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for r in rows:
        k = short(r["Kernel_Name"], 70)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    names = sorted({r["Counter_Name"] for r in rows})
    out.append(f"## PMC counters, summed over dispatches ({os.path.basename(path)})")
    for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
        out.append(f"{k}  dispatches={len(calls[k])}")
        for n in names:
            if n in agg[k]:
                out.append(f"    {n:32s} {agg[k][n]:18.0f}   per-dispatch {agg[k][n] / len(calls[k]):16.1f}")


def main():
    out = []
    for d in sys.argv[1:]:
        out.append(f"# {d}")
        for p in sorted(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)):
            kernel_stats(p, out)
        for p in sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)):
            kernel_trace(p, out)
        for p in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
            counters(p, out)
        out.append("")
    print("\n".join(out))


if __name__ == "__main__":
    main()
