#!/usr/bin/env python3
"""Per-workgroup timeline of one launch of the 1x1 GEMM form (lab build, LAB bit 128: s_memrealtime stamps at workgroup start,
after the prologue, after the k-loop, at the end + HW_ID): how many workgroups a CU holds over time, how long a slot stays
empty between two workgroups, how long each phase takes.

    SPK_LAB_LIB=tools/_bin/libspk_hip_g2lab.so python tools/lab_gemm2_timeline.py Cin Cout H cfg [mask=128]"""
import importlib
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("speak-hack_amd")
if os.environ.get("SPK_LAB_LIB"):
    pkg._lib.LIB_PATH = os.path.abspath(os.environ["SPK_LAB_LIB"])
ops = pkg.ops
Cin, Cout, H, cfg = (int(v) for v in sys.argv[1:5])
mask = int(sys.argv[5]) if len(sys.argv) > 5 else 128
dev, B, G = torch.device("cuda:0"), 8, 6
x = torch.randn(B, G * Cin, H, H, device=dev)
ws = [torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05 for _ in range(G)]
wp = ops.pack_conv_weights_list(ws, cfg)
sc = torch.rand(G * Cin, device=dev) + 0.5
sh = torch.randn(G * Cin, device=dev) * 0.1
y = torch.empty(B, G * Cout, H, H, device=dev)
stats = torch.zeros(ops.stats_slots(cfg, 1, 1, B, Cin, Cout, H, H) * 2 * G * Cout, device=dev, dtype=torch.float64)
co_t = 128 if cfg == 14 else 64
n_wg = (-(-Cout // co_t)) * (-(-B * H * H // 128)) * G
dbg = torch.zeros(n_wg * 6, device=dev, dtype=torch.int64)
os.environ["SPK_G2_LAB"] = str(mask)
for it in range(4):
    if it == 3:
        os.environ["SPK_G2_DBG"] = str(dbg.data_ptr())
    ops.conv2d_fused(x, wp, Cout, 1, 1, in_affine=(sc, sh), stats=stats, config=cfg, groups=G, out=y)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(n_wg, 6)
t0 = d[:, 0].min()
st, pro, loop, en = [(d[:, i] - t0) / 100.0 for i in range(4)]     # 100 MHz -> us
hw, xcc = d[:, 4], d[:, 5] & 0xf
cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
print(f"{Cin}->{Cout} @{H}^2 cfg {cfg} mask {mask}: {n_wg} workgroups on {len(set(cu.tolist()))} CUs; kernel span {en.max():.1f} us")
print(f"  workgroup lifetime  mean {np.mean(en - st):6.2f} us  (p10 {np.percentile(en - st, 10):.2f}, p90 {np.percentile(en - st, 90):.2f})")
print(f"  prologue            mean {np.mean(pro - st):6.2f} us   k-loop mean {np.mean(loop - pro):6.2f} us   epilogue mean {np.mean(en - loop):6.2f} us")
print(f"  first start {st.min():.2f}, last first-round start {np.sort(st)[min(767, n_wg - 1)]:.2f}, last start {st.max():.2f}, first end {en.min():.2f}")
per = defaultdict(list)
for i in range(n_wg):
    per[int(cu[i])].append((st[i], en[i]))
gaps, counts, resid = [], [], []
for k, v in per.items():
    v.sort()
    counts.append(len(v))
    # residency integral
    resid.append(sum(e - s for s, e in v) / en.max())
    ends = sorted(e for _, e in v)
    starts = sorted(s for s, _ in v)
    # a slot is refilled when a workgroup ends: pair the j-th end with the (j + resident)-th start
    nres = sum(1 for s in starts if s < ends[0])
    for j, e in enumerate(ends):
        if j + nres < len(starts):
            gaps.append(starts[j + nres] - e)
print(f"  workgroups per CU: min {min(counts)} max {max(counts)};  mean resident workgroups per CU over the span: {np.mean(resid):.2f}")
if gaps:
    print(f"  slot refill gap (end -> next start on that CU): mean {np.mean(gaps):.2f} us  p50 {np.percentile(gaps, 50):.2f}  p90 {np.percentile(gaps, 90):.2f}")
# chip-wide resident workgroups over time
ts = np.linspace(0, en.max(), 41)
occ = [(np.sum((st <= t) & (en > t))) for t in ts]
print("  resident workgroups over time: " + " ".join(f"{o}" for o in occ))
k0 = next(iter(per))
print("  one CU's workgroups (start, prologue end, loop end, end):")
for i in np.argsort(st):
    if int(cu[i]) == k0:
        print(f"     wg {i:5d}: {st[i]:7.2f} {pro[i]:7.2f} {loop[i]:7.2f} {en[i]:7.2f}")
