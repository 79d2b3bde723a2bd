#!/bin/bash
L=gpurun_out/ab_wgblocks.log; rm -f $L
for rep in 1 2; do for m in swin_s vit_s16; do for b in 512 400 250; do
  echo -n "$m VTX_WGRAD_BLOCKS=$b : " >> $L
  VTX_WGRAD_BLOCKS=$b python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $L
done; done; done
cat $L
