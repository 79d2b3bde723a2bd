#!/bin/bash
# stochastic-depth compaction threshold (% of dropped (sample, branch) pairs a layer needs to run compacted), same box
L=gpurun_out/ab_thresh.log; rm -f $L
for rep in 1 2; do for t in 2 6 10 14 100; do
  echo -n "swin_s VTX_DP_COMPACT_MIN=$t : " >> $L
  VTX_DP_COMPACT_MIN=$t python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $L
done; done
cat $L
