#!/usr/bin/env python3
"""Per-kernel PMC sums from a rocprofv3 --pmc result database: counters per dispatch, the clock the chip held
(GRBM_GUI_ACTIVE / 8 XCDs / duration) and the matrix pipe's busy share.

    python tools/pmc_summary.py x_results.db [--like gemm2]"""
import argparse
import collections
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--like", default="")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), avg(duration), lds_block_size, vgpr_count, "
                     "accum_vgpr_count from counters_collection where kernel_name like ? group by kernel_name, counter_name",
                     ("%" + a.like + "%",)).fetchall()
    d = collections.defaultdict(dict)
    for k, cn, v, n, dur, lds, vg, ag in rows:
        d[k][cn] = (v / n, dur, lds, vg, ag, n)
    for k in sorted(d):
        x = d[k]
        any_ = next(iter(x.values()))
        dur = any_[1]
        print(f"{k[:150]}\n   dispatches {any_[5]}  avg {dur / 1e3:.1f} us  lds {any_[2]}  vgpr {any_[3]}  agpr {any_[4]}")
        for cn, t in sorted(x.items()):
            print(f"     {cn:28s} {t[0]:16.0f}")
        if "GRBM_GUI_ACTIVE" in x:
            g = x["GRBM_GUI_ACTIVE"][0]
            line = f"     clock ~ {g / 8 / dur * 1e3:.0f} MHz"
            if "SQ_VALU_MFMA_BUSY_CYCLES" in x:
                line += f";  MFMA busy / (1024 SIMDs x cycles) = {x['SQ_VALU_MFMA_BUSY_CYCLES'][0] / (g / 8 * 1024):.3f}"
            print(line)


if __name__ == "__main__":
    main()
