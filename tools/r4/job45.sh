#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job45; mkdir -p $O
for w in 1 0; do
  for side in 1 0; do
  (cd /tmp && VTX_WGRAD_WIDE=$w VTX_SIDE_WGRAD=$side timeout 600 rocprofv3 --kernel-trace --stats -d $O/w$w$side -o trace -- python $R/bench.py --model swin_s --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-events > $O/run_w$w$side.log 2>&1)
  grep '"metric"' $O/run_w$w$side.log | cut -c1-120
  python tools/rocpd_stats.py $O/w$w$side/trace_results.db --steps 7 --top 40 > $O/stats_w${w}_side$side.md
  rm -rf $O/w$w$side
  grep -i "wgrad\|layer_reduce\|Total" $O/stats_w${w}_side$side.md | cut -c1-140
  done
done
