#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job27; mkdir -p $O
for rows in 21756 19600 17640 15680; do timeout 200 python tools/r4/astat_check.py --rows $rows --iters 20 2>&1 | grep -E "^M =|qkv fwd|fc1 fwd|proj|fc2 dgrad" | cut -c1-118; done > $O/rows.log 2>&1
cat $O/rows.log
for m in swin_s vit_s16; do
  timeout 600 python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' > $O/bench_$m.log
  echo "$m: $(python -c "import json;d=json.loads(open('$O/bench_$m.log').read());print(d['value'], d['ms_per_step'])")"
done
