#!/bin/bash
# HBM traffic (FETCH_SIZE x2, WRITE_SIZE) of the GEMM micro-benchmark, per dispatch in launch order.
R=$PWD; S=${1:-3}; W=${2:-fwd,dgrad,wgrad}
export TMPDIR=/tmp
O=$R/gpurun_out/pmc_gemm
mkdir -p $O; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $O -o $c -- python $R/tools/bench_gemm.py --stages $S --what $W --iters 1 > $O/$c.log 2>&1
done
python - <<PY
import sqlite3
def load(db, counter):
    c = sqlite3.connect(db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pe = [x for x in t if "pmc_event" in x][0]; pi = [x for x in t if "info_pmc" in x][0]
    kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
    q = (f"select d.dispatch_id, s.kernel_name, e.value from {pe} e join {pi} i on e.pmc_id = i.id join {kd} d on e.event_id = d.event_id "
         f"join {ks} s on d.kernel_id = s.id where i.name = ? order by d.dispatch_id")
    return list(c.execute(q, (counter,)))
f = load("$O/FETCH_SIZE_results.db", "FETCH_SIZE"); w = load("$O/WRITE_SIZE_results.db", "WRITE_SIZE")
for (i, k, fv), (_, k2, wv) in zip(f, w):
    if "gemm" in k or "wgrad" in k or "slab" in k:
        print(f"{i:5d} {k[:60]:60s} fetch(x2) {2*fv*1024/1e6:8.2f} MB  write {wv*1024/1e6:8.2f} MB")
PY
rm -f $O/*_results.db
