#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job29; mkdir -p $O
{ timeout 300 python tools/r4/astat_check.py --iters 30; timeout 300 python tools/r4/astat_check.py --vit --iters 30; timeout 300 python tools/r4/astat_check.py --rows 19600 --iters 30; } 2>&1 | grep -v "^\[W\|amdgpu.ids" > $O/astat_microbench.log
tools/battery.sh bench > $O/bench_lines.log 2>&1
timeout 900 python bench.py --model twins_svt_s --steps 20 --warmup 5 2>&1 | grep '"metric"' > gpurun_out/bench_twins_svt_s.log
cut -c1-200 $O/bench_lines.log; cat $O/astat_microbench.log
