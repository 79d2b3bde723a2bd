"""For every dispatch of a kernel (name substring) in a rocprofv3 rocpd trace: how much of its duration another kernel was running too.
    python tools/r6/overlap_of.py x_results.db oneRankReduce"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else None
rows = c.execute(f"select s.kernel_name, d.start, d.end{', d.' + qcol if qcol else ''} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
hits = [r for r in rows if pat in r[0]]
print(f"{len(hits)} dispatches of *{pat}*; columns of the dispatch table: {cols}")
tot = ov = 0.0
queues = {}
for r in rows:
    if qcol: queues.setdefault(r[3], [0, set()]); queues[r[3]][0] += 1; queues[r[3]][1].add(r[0][:40])
for h in hits[len(hits) // 2:]:
    d = h[2] - h[1]; o = 0
    for r in rows:
        if r is h or r[2] <= h[1] or r[1] >= h[2]: continue
        o = max(o, min(r[2], h[2]) - max(r[1], h[1]))
    tot += d; ov += o
print(f"second half of them: {tot / 1e3:.1f} us in total, {ov / 1e3:.1f} us of it with another kernel running ({100 * ov / max(tot, 1):.0f} %)")
if qcol:
    for q, (n, names) in sorted(queues.items()): print(f"queue {q}: {n} dispatches, e.g. {sorted(names)[:4]}")
