#!/bin/bash
# Round-5 profile battery without the tests / micro-benchmarks (tools/battery.sh prof runs those too): rocprofv3 kernel stats of the five
# workloads (single stream) + PMC traffic of three.  Run through gpurun from the repo root.
R=$PWD; export TMPDIR=/tmp; TAG=${1:-r5}
tools/gpu_check.sh prof:$TAG profvit:$TAG profpvt:$TAG
for m in dino twins_svt_s; do
  mkdir -p gpurun_out/prof${m}_$TAG
  (cd /tmp && VTX_SIDE_WGRAD=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof${m}_$TAG -o trace -- \
     python $R/bench.py --model $m --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/prof${m}_$TAG/run.log 2>&1)
  python tools/rocpd_stats.py gpurun_out/prof${m}_$TAG/trace_results.db --steps 5 --top 70 > gpurun_out/prof${m}_$TAG/kernel_stats.md
  rm -f gpurun_out/prof${m}_$TAG/trace_results.db
done
for m in swin_s vit_s16 pvt_small; do timeout 1500 tools/pmc_traffic.sh $m > /dev/null 2>&1; done
head -12 gpurun_out/prof_$TAG/kernel_stats.md | cut -c1-150
