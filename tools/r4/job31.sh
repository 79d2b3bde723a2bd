#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job31; mkdir -p $O
run() { timeout 600 python bench.py --model $1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' > $O/b.log; python -c "import json;d=json.loads(open('$O/b.log').read());print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for t in 8 4 2 0; do echo "swin_s VTX_DP_COMPACT_MIN=$t: $(VTX_DP_COMPACT_MIN=$t run swin_s)"; done
  echo "vit_s16 default: $(run vit_s16)"
  echo "vit_s16 compaction forced: $(VTX_VIT_COMPACT=1 VTX_DP_COMPACT_MIN=4 run vit_s16)"
done
