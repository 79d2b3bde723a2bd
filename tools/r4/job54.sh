#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/proftwins_r4w; mkdir -p $O
(cd /tmp && VTX_SIDE_WGRAD=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O -o trace -- python $R/bench.py --model twins_svt_s --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/run.log 2>&1)
grep '"metric"' $O/run.log | cut -c1-160
python tools/rocpd_stats.py $O/trace_results.db --steps 7 --top 60 > $O/kernel_stats.md
rm -f $O/trace_results.db
head -14 $O/kernel_stats.md | cut -c1-150
