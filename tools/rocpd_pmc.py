#!/usr/bin/env python3
"""Per-kernel PMC counter summary from a rocprofv3 --pmc rocpd database (mean per dispatch).

    python tools/rocpd_pmc.py gpurun_out/pmc/x_results.db [--match substring]
"""
import argparse
import sqlite3
import subprocess


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--match", default="")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pe = [x for x in t if "pmc_event" in x][0]
    pi = [x for x in t if "info_pmc" in x][0]
    kd = [x for x in t if "kernel_dispatch" in x][0]
    ks = [x for x in t if "kernel_symbol" in x][0]
    q = (f"select s.kernel_name, i.name, count(*), avg(e.value) from {pe} e join {pi} i on e.pmc_id = i.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, i.name")
    rows = c.execute(q).fetchall()
    names = sorted({r[0] for r in rows})
    try:
        dm = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(n.replace(".kd", "").replace("DF16b", "u6__bf16") for n in names),
                                            capture_output=True, text=True).stdout.split("\n")))
    except Exception:
        dm = {n: n for n in names}
    per = {}
    for k, cn, n, v in rows:
        per.setdefault(k, {})[cn] = (n, v)
    for k in names:
        d = dm.get(k, k)
        if a.match and a.match not in d:
            continue
        print(f"## {d[:100]}  ({next(iter(per[k].values()))[0]} dispatches)")
        for cn in sorted(per[k]):
            print(f"   {cn:32s} {per[k][cn][1]:16.1f}")


if __name__ == "__main__":
    main()
