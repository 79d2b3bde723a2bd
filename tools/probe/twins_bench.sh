#!/bin/bash
# Twins-SVT-S (the row after F1-F4): bench line + rocprofv3 kernel stats (single-stream, attributable durations).
R=$PWD
export TMPDIR=/tmp
timeout 700 python bench.py --model twins_svt_s --steps 20 --warmup 5 2>&1 | grep '"metric"' > gpurun_out/bench_twins_svt_s.log
mkdir -p gpurun_out/proftwins_r3
(cd /tmp && VTX_SIDE_WGRAD=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/proftwins_r3 -o trace -- \
   python $R/bench.py --model twins_svt_s --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/proftwins_r3/run.log 2>&1)
python tools/rocpd_stats.py gpurun_out/proftwins_r3/trace_results.db --steps 7 --top 40 > gpurun_out/proftwins_r3/kernel_stats.md
rm -f gpurun_out/proftwins_r3/trace_results.db
cut -c1-400 gpurun_out/bench_twins_svt_s.log
head -30 gpurun_out/proftwins_r3/kernel_stats.md | cut -c1-150
