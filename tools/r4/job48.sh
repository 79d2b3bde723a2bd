#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job48; mkdir -p $O
for m in swin_s vit_s16; do
for side in 1 0; do
  (cd /tmp && VTX_SIDE_WGRAD=$side timeout 600 rocprofv3 --kernel-trace -d $O/t -o trace -- python $R/bench.py --model $m --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-events > $O/run_${m}_$side.log 2>&1)
  echo "== $m side=$side"; grep '"metric"' $O/run_${m}_$side.log | cut -c1-110
  python tools/rocpd_gaps.py $O/t/trace_results.db --skip 0.6 2>&1 | head -30
  rm -rf $O/t
done
done > $O/gaps.log 2>&1
cat $O/gaps.log | cut -c1-170
