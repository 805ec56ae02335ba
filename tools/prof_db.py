#!/usr/bin/env python3
"""Per-kernel table from a rocprofv3 result database (``rocprofv3 --kernel-trace -d DIR -o NAME`` writes NAME_results.db on this
image): calls, total / average duration, share -- the text committed under profiles/.

    python tools/prof_db.py gpurun_out/.../x_results.db [--top 40] [--like PATTERN] [--by-grid] [--by-stream]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--like", default=None, help="only kernels whose name contains this")
    ap.add_argument("--by-grid", action="store_true", help="split every kernel by launch grid")
    ap.add_argument("--by-stream", action="store_true", help="one table per HIP stream (which kernels sit on the critical stream)")
    args = ap.parse_args()
    c = sqlite3.connect(args.db)
    if args.by_stream:
        streams = c.execute("select stream, count(*), sum(end-start) from kernels group by stream order by sum(end-start) desc").fetchall()
        for st, n, tot in streams:
            print(f"== {st}: {n} dispatches, {tot / 1e6:.3f} ms of kernel time")
            print(f"{'total ms':>10s} {'share':>6s} {'calls':>7s} {'avg us':>9s}  kernel")
            rows = c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels where stream = ? group by name "
                             "order by sum(end-start) desc", (st,)).fetchall()
            for name, cnt, t, avg in rows[:args.top]:
                print(f"{t / 1e6:10.3f} {100 * t / tot:5.1f}% {cnt:7d} {avg / 1e3:9.1f}  {name[:150]}")
        return
    where = "where name like ?" if args.like else ""
    par = ("%" + args.like + "%",) if args.like else ()
    grp = "name, grid_x, grid_y, grid_z" if args.by_grid else "name"
    rows = c.execute(f"select {grp}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels {where} "
                     f"group by {grp} order by sum(end-start) desc", par).fetchall()
    total = sum(r[-4] for r in rows) or 1
    n = sum(r[-5] for r in rows)
    print(f"{n} dispatches, {total / 1e6:.3f} ms of kernel time, {len(rows)} rows")
    print(f"{'total ms':>10s} {'share':>6s} {'calls':>7s} {'avg us':>9s} {'min us':>9s} {'max us':>9s}  kernel")
    for r in rows[:args.top]:
        name = r[0] if not args.by_grid else f"{r[0][:110]}  grid=({r[1]},{r[2]},{r[3]})"
        cnt, tot, avg, mn, mx = r[-5:]
        print(f"{tot / 1e6:10.3f} {100 * tot / total:5.1f}% {cnt:7d} {avg / 1e3:9.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f}  {name[:170]}")


if __name__ == "__main__":
    main()
