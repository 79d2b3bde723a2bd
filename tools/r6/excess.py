"""Where a step's time sits above what its launches' algorithmic bytes / FLOPs need: reads a `bench.py --shape-table` file and ranks the
(kernel, shape) rows by excess = ms/step - max(bytes / 4.5 TB/s, flops / 700 TFLOP/s) (the streaming rate and the GEMM rate the best
launches of this build reach on this chip: a ranking device, not a roofline).   python tools/r6/excess.py table.md [top]"""
import re, sys
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\| `(.+?)` \| (.+?) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \| (\d+) \| (\d+) \|", l)
    if not m:
        continue
    k, shp, n, us, ms, gbs, tf = m.group(1), m.group(2), float(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6)), float(m.group(7))
    byt, fl = gbs * 1e9 * ms * 1e-3, tf * 1e12 * ms * 1e-3
    floor = max(byt / 4.5e12, fl / 700e12) * 1e3
    rows.append((ms - floor, k, shp, n, us, ms, gbs, tf, floor))
tot = sum(r[5] for r in rows)
print(f"{len(rows)} rows, {tot:.3f} ms/step in library launches; excess over max(bytes / 4.5 TB/s, flops / 700 TFLOP/s): {sum(r[0] for r in rows):.3f} ms")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
print("| excess ms | kernel | shape | launches | avg us | ms/step | GB/s | TFLOP/s |\n|---|---|---|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -r[0])[:top]:
    print(f"| {r[0]:.3f} | `{r[1]}` | {r[2]} | {r[3]:.0f} | {r[4]:.1f} | {r[5]:.3f} | {r[6]:.0f} | {r[7]:.0f} |")
# by kernel family
fam = {}
for r in rows:
    f = r[1].split("<")[0]
    a = fam.setdefault(f, [0.0, 0.0])
    a[0] += r[5]; a[1] += r[0]
print("\n| family | ms/step | excess ms |\n|---|---|---|")
for f, (a, b) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{f}` | {a:.3f} | {b:.3f} |")
