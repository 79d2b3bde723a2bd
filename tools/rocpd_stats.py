#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats table (like --stats CSV).

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--steps N] [--top 40] > profiles/round1_xxx.md
"""
import argparse
import re
import sqlite3
import subprocess


def demangle(names):
    bare = [(n[:-3] if n.endswith(".kd") else n).replace("DF16b", "u6__bf16") for n in names]   # GNU c++filt lacks __bf16      # rocprofv3 appends ".kd" (kernel descriptor)
    try:
        out = subprocess.run(["c++filt"], input="\n".join(bare), capture_output=True,
                             text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def short(n):
    n = n.replace("void ", "", 1).replace("(anonymous namespace)::", "")      # (kernels of csrc/gemm_pp.hip: unnamed namespace)
    depth = 0
    for i, ch in enumerate(n):              # drop the argument list (the first "(" outside template brackets)
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            n = n[:i]
            break
    return n[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=0, help="divide totals by this many steps")
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--neighbours", default="", help="kernel-name substring: print which kernels run right before / after it")
    ap.add_argument("--by-grid", default="", help="kernel-name substring: split that kernel's rows by launch grid (one row per shape)")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if "kernel_dispatch" in x][0]
    ks = [x for x in t if "kernel_symbol" in x][0]
    rows = c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                     f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name").fetchall()
    dm = demangle([r[0] for r in rows])
    if a.by_grid:
        cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
        gx, gy = [x for x in cols if "grid" in x and x.endswith("x")][0], [x for x in cols if "grid" in x and x.endswith("y")][0]
        wx = [x for x in cols if "workgroup" in x and x.endswith("x")][0]
        q = c.execute(f"select s.kernel_name, d.{gx}, d.{gy}, d.{wx}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                      f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name, d.{gx}, d.{gy}, d.{wx}").fetchall()
        q = [r for r in q if a.by_grid in short(dm[r[0]])]
        q.sort(key=lambda r: -r[5])
        div = a.steps if a.steps else 1
        print(f"| kernel | grid (workgroups x, y) | calls | total ms{' /step' if a.steps else ''} | avg us | min us | max us |")
        print("|---|---|---|---|---|---|---|")
        for name, x, y, w, n, t, mn, mx in q:
            print(f"| `{short(dm[name])[:60]}` | {x // max(w, 1)} x {y} | {n} | {t / 1e6 / div:.3f} | {t / n / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} |")
        return
    if a.neighbours:
        seq = c.execute(f"select s.kernel_name, d.start from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
        names = [short(dm[n]) for n, _ in seq]
        from collections import Counter
        cnt = Counter()
        for i, n in enumerate(names):
            if a.neighbours in n:
                cnt[(names[i - 1][:70] if i else "-", names[i + 1][:70] if i + 1 < len(names) else "-")] += 1
        print(f"## neighbours of *{a.neighbours}*: count | previous kernel | next kernel")
        for (p, q), k in cnt.most_common(25):
            print(f"{k:6d} | {p} | {q}")
        return
    tot = sum(r[2] for r in rows)
    rows.sort(key=lambda r: -r[2])
    div = a.steps if a.steps else 1
    print(f"| kernel | calls | total ms{' /step' if a.steps else ''} | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, n, s, mn, mx in rows[:a.top]:
        print(f"| `{short(dm[name])}` | {n} | {s / 1e6 / div:.3f} | {s / n / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | "
              f"{100 * s / tot:.1f} |")
    print(f"\nTotal GPU kernel time: {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches"
          + (f" = {tot / 1e6 / div:.2f} ms/step" if a.steps else ""))


if __name__ == "__main__":
    main()
