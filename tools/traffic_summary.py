#!/usr/bin/env python3
"""HBM traffic of the decoder step's conv launches from two rocprofv3 PMC passes over tools/pmc_decoder.py
(--pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs, each with --kernel-trace).

    python tools/traffic_summary.py gpurun_out/r2/pmc_fetch gpurun_out/r2/pmc_write [--json profiles/conv_traffic.json]

Calls are told apart by their prologue launch (the first kernel of a decoder forward); the first call of each precision builds
the launch plan and is dropped.  FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md, HBM section).
"""
import csv
import glob
import json
import sys
from collections import defaultdict


UPS = {}


def calls(d, counter):
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out, cur = [], []
    for r in rows:
        n = r["Kernel_Name"]
        if "conv_kernel<" in n and ", 3, 3, 1, " in n or "conv3x3_bf16x3_kernel" in n or "wino_kernel" in n:
            cur.append((n, float(r["Counter_Value"]) * 1024.0))
        if "upsample2x_" in n and "bwd" not in n:      # the x2 image a Winograd x2 layer reads (its own launch): reported beside the convs
            UPS.setdefault(counter, []).append(float(r["Counter_Value"]) * 1024.0)
        if "bias_noise_style_kernel" in n and cur:     # a forward's FIRST launch (the const-input prologue): the previous one is complete
            out.append(cur)                            # (its last launch used to be toRGB -- now inside the last conv's epilogue)
            cur = []
    if cur:
        out.append(cur)
    return out


def flat_total(d, counter):
    """All 3x3 stride-1 conv launches of a run: (bytes, dispatches)."""
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "conv_kernel<" in r["Kernel_Name"]
            and ", 3, 3, 1, " in r["Kernel_Name"]]
    return sum(float(r["Counter_Value"]) * 1024.0 for r in rows), len(rows)


def main():
    if "--sg2" in sys.argv:
        # the StyleGAN2 variant's step (tools/pmc_decoder.py --sg2: N forwards, the first builds the launch plan): its skip toRGB
        # launches do not mark a forward's end, so the conv launches are averaged over all forwards of the run
        fb, nf = flat_total(sys.argv[1], "FETCH_SIZE")
        wb, nw = flat_total(sys.argv[2], "WRITE_SIZE")
        print(f"sg2    : {nf} conv dispatches; FETCH {fb / nf / 1e6:8.1f} MB, WRITE {wb / nw / 1e6:8.1f} MB per conv launch "
              f"= {(fb / nf + wb / nw) / 1e6:6.1f} MB")
        if "--json" in sys.argv:
            path = sys.argv[sys.argv.index("--json") + 1]
            out = json.load(open(path))
            out["sg2_bytes_per_launch"] = int(fb / nf + wb / nw)
            json.dump(out, open(path, "w"), indent=1)
        return
    fetch, write = calls(sys.argv[1], "FETCH_SIZE"), calls(sys.argv[2], "WRITE_SIZE")
    res = {}
    for tag, pred in (("f32", lambda c: not any("bf16x3" in n for n, _ in c)), ("bf16x3", lambda c: any("bf16x3" in n for n, _ in c))):
        fc = [c for c in fetch if pred(c) and len(c) == 12][1:]
        wc = [c for c in write if pred(c) and len(c) == 12][1:]
        fb = sum(v for c in fc for _, v in c) / len(fc)
        wb = sum(v for c in wc for _, v in c) / len(wc)
        res[tag] = (fb, wb, len(fc))
        print(f"{tag:7s}: {len(fc)} steps counted; per step: FETCH {fb / 1e6:8.1f} MB, WRITE {wb / 1e6:8.1f} MB, "
              f"sum {(fb + wb) / 1e6:8.1f} MB = {(fb + wb) / 12 / 1e6:6.1f} MB per conv launch")
        if tag == "f32" and UPS:
            n_all = sum(1 for c in fetch if len(c) == 12)
            print(f"         + the upsample2x launches of the Winograd x2 layers (all {n_all} forwards of the run, both precisions): "
                  f"FETCH {sum(UPS.get('FETCH_SIZE', [])) / n_all / 1e6:8.1f} MB, WRITE {sum(UPS.get('WRITE_SIZE', [])) / n_all / 1e6:8.1f} MB per forward")
        per = defaultdict(lambda: [0.0, 0.0, 0])
        for c in fc:
            for n, v in c:
                per[n][0] += v / len(fc)
                per[n][2] += 1
        for c in wc:
            for n, v in c:
                per[n][1] += v / len(wc)
        for n, (f, w, k) in sorted(per.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
            print(f"    {n.replace('void ', '')[:86]:86s} launches/step {k // len(fc):2d}  fetch {f / 1e6:8.1f} MB  write {w / 1e6:8.1f} MB")
    if "--json" in sys.argv:
        fb, wb, n = res["f32"]
        out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 tools/pmc_decoder.py` "
                         "(the headline decoder step launched eagerly through its launch plan); summary in profiles/r04_g_conv_traffic_final.txt",
               "kernel": "spkwino::wino_kernel (10 launches: 8 with 32 x 8 regions -- the last with toRGB in its epilogue --, 2 with 16 x 16 regions and a sliced contraction) + spkconv::conv_kernel<Cfg,3,3,1,MODE 0|1> (the two 8^2 layers)", "launches_per_step": 12, "steps_counted": n,
               "fetch_bytes_per_step": int(fb), "write_bytes_per_step": int(wb), "bytes_per_launch": int((fb + wb) / 12),
               "bf16x3_fetch_bytes_per_step": int(res["bf16x3"][0]), "bf16x3_write_bytes_per_step": int(res["bf16x3"][1]),
               "correction": "FETCH_SIZE / WRITE_SIZE are KiB.  The input gathers are 4-byte-per-lane loads, which FETCH_SIZE counts in full "
                             "(calibrated in round 1 on the 64->64@256^2 layer: 1.37x the input bytes = the (10*34)/(8*32) halo factor + "
                             "weights); the 16-byte weight loads are L2 hits; the gfx950 x2 correction for 16-B/lane streams is therefore NOT "
                             "applied.  WRITE_SIZE = output bytes + the split-K partial slabs of the <= 32^2 layers."}
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
