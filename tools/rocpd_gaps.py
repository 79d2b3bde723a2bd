#!/usr/bin/env python3
"""Idle time BETWEEN kernels in a rocprofv3 (rocpd sqlite) kernel trace: is the GPU ever without a kernel inside a train step?

    python tools/rocpd_gaps.py x_results.db [--skip 0.5]

Takes the dispatches of the last (1 - skip) part of the trace (steady state), merges their [start, end] intervals (two
streams overlap), and reports wall, busy (union), idle = wall - busy, the concurrency (sum of durations / busy), a
histogram of the gaps and which kernel precedes the gaps that add up to most.
"""
import argparse
import sqlite3
from collections import defaultdict

from rocpd_stats import demangle, short


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--skip", type=float, default=0.5, help="leading fraction of the dispatches to ignore (start-up, warm-up)")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if "kernel_dispatch" in x][0]
    ks = [x for x in t if "kernel_symbol" in x][0]
    rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    rows = rows[int(len(rows) * a.skip):]
    dm = demangle(sorted({r[0] for r in rows}))
    wall = max(r[2] for r in rows) - rows[0][1]
    busy = 0
    total = sum(r[2] - r[1] for r in rows)
    gaps = []                                   # (length, name of the kernel that ended last before the gap)
    cur_s, cur_e, last = rows[0][1], rows[0][2], rows[0][0]
    for name, s, e in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last, name))
            cur_s, cur_e, last = s, e, name
        elif e > cur_e:
            cur_e, last = e, name
    busy += cur_e - cur_s
    print(f"dispatches {len(rows)}  wall {wall / 1e6:.2f} ms  busy (union) {busy / 1e6:.2f} ms  idle {(wall - busy) / 1e6:.2f} ms "
          f"({100 * (wall - busy) / wall:.1f} %)  sum of durations {total / 1e6:.2f} ms  concurrency {total / busy:.3f}")
    edges = [0, 1, 2, 4, 8, 16, 50, 1e9]
    print("gap histogram (us): count, total ms")
    for lo, hi in zip(edges, edges[1:]):
        g = [x[0] for x in gaps if lo * 1e3 <= x[0] < hi * 1e3]
        print(f"  [{lo:g}, {hi:g}) {len(g):6d} {sum(g) / 1e6:8.3f}")
    by = defaultdict(lambda: [0, 0])
    for g, prev, nxt in gaps:
        k = (short(dm[prev])[:50], short(dm[nxt])[:50])
        by[k][0] += 1
        by[k][1] += g
    print("largest gap totals: count | total ms | avg us | kernel before -> kernel after")
    for k, (n, s) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{n:6d} | {s / 1e6:7.3f} | {s / n / 1e3:6.2f} | {k[0]} -> {k[1]}")


if __name__ == "__main__":
    main()
