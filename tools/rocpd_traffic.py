#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc rocpd databases (FETCH_SIZE pass, WRITE_SIZE pass).

FETCH_SIZE / WRITE_SIZE are reported in KiB-units of 1024 B (rocprofv3 derived counters over TCC_EA0_RDREQ /
TCC_EA0_WRREQ).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-byte requests of wide
coalesced reads (16 B/lane global_load and LDS-DMA alike) at 64 B, i.e. reports exactly half -> doubled here
(`fetch_x2`); WRITE_SIZE is taken as reported (uncalibrated in the guide).  Values are means per dispatch.
"""
import argparse
import json
import sqlite3
import subprocess


def load(db, counter):
    c = sqlite3.connect(db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pe = [x for x in t if "pmc_event" in x][0]
    pi = [x for x in t if "info_pmc" in x][0]
    kd = [x for x in t if "kernel_dispatch" in x][0]
    ks = [x for x in t if "kernel_symbol" in x][0]
    q = (f"select s.kernel_name, count(*), avg(e.value), sum(e.value) from {pe} e join {pi} i on e.pmc_id = i.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id where i.name = ? "
         f"group by s.kernel_name")
    return {k: (n, avg, tot) for k, n, avg, tot in c.execute(q, (counter,))}


def short(name):
    """'void gemm_glds_kernel<128, 128, 64, 2>(GemmArgs)' -> 'gemm_glds_kernel<128, 128, 64, 2>' (template args kept)."""
    if name.startswith("void "):
        name = name[5:]
    name = name.replace("(anonymous namespace)::", "")      # (kernels of csrc/gemm_pp.hip live in an unnamed namespace)
    depth = 0
    for i, ch in enumerate(name):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return name[:i]
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_db")
    ap.add_argument("write_db")
    ap.add_argument("--model", default="swin_s")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    f = load(a.fetch_db, "FETCH_SIZE")
    w = load(a.write_db, "WRITE_SIZE")
    names = sorted(set(f) | set(w))
    bare = [(n[:-3] if n.endswith(".kd") else n).replace("DF16b", "u6__bf16") for n in names]   # GNU c++filt lacks __bf16
    try:
        dem = subprocess.run(["c++filt"], input="\n".join(bare), capture_output=True,
                             text=True).stdout.split("\n")
        dm = dict(zip(names, dem))
    except Exception:
        dm = dict(zip(names, bare))
    rows = []
    for k in names:
        n, favg, ftot = f.get(k, (0, 0.0, 0.0))
        _, wavg, wtot = w.get(k, (0, 0.0, 0.0))
        rows.append(dict(kernel=short(dm.get(k, k)), dispatches=n, fetch_x2_bytes=2 * favg * 1024,
                         write_bytes=wavg * 1024, total_bytes=(2 * favg + wavg) * 1024,
                         sum_bytes=(2 * ftot + wtot) * 1024))
    rows.sort(key=lambda r: -r["sum_bytes"])
    print(f"# HBM traffic per kernel, {a.model} bench (PMC: FETCH_SIZE x2 [gfx950 correction], WRITE_SIZE), mean per dispatch\n")
    print("| kernel | dispatches | fetch MB (x2) | write MB | total MB / dispatch | share of all traffic |")
    print("|---|---|---|---|---|---|")
    allb = sum(r["sum_bytes"] for r in rows) or 1.0
    for r in rows[:40]:
        print(f"| `{r['kernel'][:90]}` | {r['dispatches']} | {r['fetch_x2_bytes'] / 1e6:.2f} | {r['write_bytes'] / 1e6:.2f} | "
              f"{r['total_bytes'] / 1e6:.2f} | {100 * r['sum_bytes'] / allb:.1f} % |")
    print(f"\nTotal over the profiled run: {allb / 1e9:.2f} GB")
    if a.json:
        json.dump({r["kernel"]: dict(fetch_x2_bytes=round(r["fetch_x2_bytes"]),
                                                   write_bytes=round(r["write_bytes"]),
                                                   dispatches=r["dispatches"]) for r in rows}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
