#!/bin/bash
R=$PWD; export TMPDIR=/tmp
tools/battery.sh bench > gpurun_out/bench_lines_final.log 2>&1
timeout 900 python bench.py --model twins_svt_s --steps 20 --warmup 5 2>&1 | grep '"metric"' > gpurun_out/bench_twins_svt_s.log
mkdir -p gpurun_out/profdino_r4b
(cd /tmp && VTX_SIDE_WGRAD=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profdino_r4b -o trace -- python $R/bench.py --model dino --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/profdino_r4b/run.log 2>&1)
python tools/rocpd_stats.py gpurun_out/profdino_r4b/trace_results.db --steps 5 --top 70 > gpurun_out/profdino_r4b/kernel_stats.md
rm -f gpurun_out/profdino_r4b/trace_results.db
cut -c1-170 gpurun_out/bench_lines_final.log; cut -c1-170 gpurun_out/bench_twins_svt_s.log
