#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job52; mkdir -p $O
for rep in 1 2; do
for th in 8 4 2 12; do
  echo -n "swin_s compact_min=$th: "
  VTX_DP_COMPACT_MIN=$th timeout 300 python bench.py --model swin_s --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
done | tee $O/thresh.log
