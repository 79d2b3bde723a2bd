#!/bin/bash
# idle time between kernels, Swin-S and ViT-S/16, default (two streams) and single stream
R=$PWD; export TMPDIR=/tmp
for m in swin_s vit_s16; do
  for side in 1 0; do
    O=$R/gpurun_out/gaps_${m}_$side; mkdir -p $O
    (cd /tmp && VTX_SIDE_WGRAD=$side timeout 600 rocprofv3 --kernel-trace -d $O -o trace -- \
       python $R/bench.py --model $m --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-events --no-secondary > $O/run.log 2>&1)
    echo "== $m VTX_SIDE_WGRAD=$side: $(grep -o '"ms_per_step": [0-9.]*' $O/run.log)"
    python $R/tools/rocpd_gaps.py $O/trace_results.db --skip 0.6
    rm -f $O/trace_results.db
  done
done
