R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/prof_copy
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_copy -o t -- python $R/tools/probe/copy_floor.py > $R/gpurun_out/prof_copy/run.log 2>&1)
python - <<PY
import sqlite3
c = sqlite3.connect("$R/gpurun_out/prof_copy/t_results.db")
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
for n, g, cnt, avg, mn in c.execute(f"select s.kernel_name, d.grid_size_x, count(*), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name, d.grid_size_x"):
    print(f"{n[:60]:60s} grid {g:9d} calls {cnt:3d} avg {avg/1e3:7.2f} us min {mn/1e3:7.2f}")
PY
rm -f $R/gpurun_out/prof_copy/t_results.db
