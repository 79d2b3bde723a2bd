#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job22; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '"metric"' > $O/bench_swin_s.json
timeout 900 python bench.py --model vit_s16 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '"metric"' > $O/bench_vit_s16.json
python - <<'PY'
import json
for m in ("swin_s", "vit_s16"):
    d = json.loads(open(f"gpurun_out/r4job22/bench_{m}.json").read())
    print(m, d["value"], d["ms_per_step"], {k: d[k] for k in d if "frac" in k or k == "roofline"})
    ks = d.get("kernels", [])
    for k in ks[:28]:
        print("   ", {a: (round(b, 3) if isinstance(b, float) else b) for a, b in k.items()})
PY
