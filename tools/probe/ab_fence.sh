#!/bin/bash
L=gpurun_out/ab_fence.log; rm -f $L
for rep in 1 2 3; do for f in 0 merge 1; do
  echo -n "swin_s VTX_SIDE_FENCE=$f : " >> $L
  VTX_SIDE_FENCE=$f python bench.py --model swin_s --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $L
done; done
cat $L
