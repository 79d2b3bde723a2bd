#!/bin/bash
# Standard GPU battery (run through gpurun from the repo root):  tools/gpu_check.sh [tests] [bench] [prof TAG]
set -u
R=$PWD
mkdir -p gpurun_out
for what in "$@"; do
  case "$what" in
    tests)
      rm -f gpurun_out/parity.log
      timeout 1500 python -m pytest tests -q -m gpu -x --timeout 900 2>&1 | tail -15 | tee gpurun_out/tests.log ;;
    bench)
      timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/bench.log ;;
    benchq)
      timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/benchq.log ;;
    profpvt:*)
      tag=${what#profpvt:}
      mkdir -p gpurun_out/profpvt_${tag}
      export TMPDIR=/tmp
      (cd /tmp && VTX_SIDE_WGRAD=0 timeout 1200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profpvt_${tag} -o trace -- \
         python $R/bench.py --model pvt_small --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/profpvt_${tag}/run.log 2>&1)
      grep '"metric"' gpurun_out/profpvt_${tag}/run.log | cut -c1-220
      python tools/rocpd_stats.py gpurun_out/profpvt_${tag}/trace_results.db --steps 7 --top 70 > gpurun_out/profpvt_${tag}/kernel_stats.md
      rm -f gpurun_out/profpvt_${tag}/trace_results.db
      head -16 gpurun_out/profpvt_${tag}/kernel_stats.md | cut -c1-170 ;;
    profvit:*)
      tag=${what#profvit:}
      mkdir -p gpurun_out/profvit_${tag}
      export TMPDIR=/tmp
      (cd /tmp && VTX_SIDE_WGRAD=0 timeout 1200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profvit_${tag} -o trace -- \
         python $R/bench.py --model vit_s16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/profvit_${tag}/run.log 2>&1)
      grep '"metric"' gpurun_out/profvit_${tag}/run.log | cut -c1-220
      python tools/rocpd_stats.py gpurun_out/profvit_${tag}/trace_results.db --steps 7 --top 70 > gpurun_out/profvit_${tag}/kernel_stats.md
      rm -f gpurun_out/profvit_${tag}/trace_results.db
      head -16 gpurun_out/profvit_${tag}/kernel_stats.md | cut -c1-170 ;;
    prof:*)
      tag=${what#prof:}
      mkdir -p gpurun_out/prof_${tag}
      export TMPDIR=/tmp
      # (single-stream: with the weight gradients on a second stream kernel durations overlap and are not attributable)
      (cd /tmp && VTX_SIDE_WGRAD=0 timeout 1200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag} -o trace -- \
         python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-secondary > $R/gpurun_out/prof_${tag}/run.log 2>&1)
      grep '"metric"' gpurun_out/prof_${tag}/run.log | cut -c1-220
      python tools/rocpd_stats.py gpurun_out/prof_${tag}/trace_results.db --steps 7 --top 70 > gpurun_out/prof_${tag}/kernel_stats.md
      python tools/rocpd_stats.py gpurun_out/prof_${tag}/trace_results.db --neighbours FillFunctor > gpurun_out/prof_${tag}/fill_neighbours.txt
      python tools/rocpd_stats.py gpurun_out/prof_${tag}/trace_results.db --neighbours copyBuffer > gpurun_out/prof_${tag}/copy_neighbours.txt
      rm -f gpurun_out/prof_${tag}/trace_results.db
      head -24 gpurun_out/prof_${tag}/kernel_stats.md | cut -c1-170 ;;
  esac
done
