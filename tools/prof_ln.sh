#!/bin/bash
# GPU-side durations of the LayerNorm kernels on the micro-benchmark shapes (the Python loop itself is host-bound).
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/prof_ln
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ln -o t -- python $R/tools/bench_ln.py > $R/gpurun_out/prof_ln/run.log 2>&1)
python - <<PY
import sqlite3
c = sqlite3.connect("$R/gpurun_out/prof_ln/t_results.db")
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
rows = c.execute(f"select s.kernel_name, d.grid_size_x, count(*), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id where s.kernel_name like '%ln_%' group by s.kernel_name, d.grid_size_x order by s.kernel_name").fetchall()
for n, g, cnt, avg, mn in rows:
    print(f"{n[:48]:48s} grid {g:8d} calls {cnt:4d} avg {avg/1e3:7.2f} us  min {mn/1e3:7.2f} us")
PY
rm -f $R/gpurun_out/prof_ln/t_results.db
