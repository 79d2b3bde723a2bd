#!/bin/bash
# Round-6 GPU jobs, one parameterised script (run through gpurun from the repo root): tools/r6/job.sh <job> [args]
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'tools/r6/job.sh tests; tools/r6/job.sh bench'
# Everything a job prints goes to gpurun_out/r6_<job>.log; profiles to gpurun_out/r6_<job>_*.
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
J=$1; shift
LOG=gpurun_out/r6_$J.log
case "$J" in
  tests)      # the GPU test suite (optionally a file first, then -k expr ...)
    if [ $# -gt 0 ] && [ -e "$1" ]; then T="$1"; shift; else T=tests; fi
    timeout 2400 python -m pytest $T -m gpu ${XFLAG:--x} -q "$@" 2>&1 | tail -40 > $LOG; tail -40 $LOG ;;
  bench)      # headline (+ secondaries unless flags say otherwise)
    timeout 1200 python bench.py --steps 20 --warmup 5 "$@" 2>&1 | grep '"metric"' > $LOG; cut -c1-600 $LOG ;;
  ab)         # same-box A/B of one option on one model: tools/r6/job.sh ab swin_s OPTION v0 v1 ...   (REPS, STEPS from the environment)
    M=$1; O=$2; shift 2
    : > $LOG
    for rep in $(seq 1 ${REPS:-2}); do for v in "$@"; do
      echo "== $O=$v" >> $LOG
      env VTX_$O=$v timeout 600 python bench.py --model $M --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' | cut -c1-200 >> $LOG
    done; done
    cat $LOG ;;
  ablib)      # same-box A/B of two library builds: tools/r6/job.sh ablib swin_s libA.so libB.so
    M=$1; shift
    : > $LOG
    for rep in $(seq 1 ${REPS:-2}); do for l in "$@"; do
      echo "== $l" >> $LOG
      env VTX_LIBVTX=$R/$l timeout 600 python bench.py --model $M --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' | cut -c1-200 >> $LOG
    done; done
    cat $LOG ;;
  envab)      # same-box A/B of an ENVIRONMENT variable: tools/r6/job.sh envab swin_s "extra bench args" VAR v0 v1 ...
    M=$1; X=$2; V=$3; shift 3
    : > $LOG
    for rep in $(seq 1 ${REPS:-2}); do for v in "$@"; do
      echo "== $V=$v" >> $LOG
      env $V=$v timeout 600 python bench.py --model $M --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events $X 2>&1 | grep '"metric"' | cut -c1-200 >> $LOG
    done; done
    cat $LOG ;;
  ddp)        # what the data-parallel machinery costs on one GPU (one-rank RCCL group, forced) vs the bypass
    timeout 1500 python tools/r6/ddp_overhead_one_gpu.py "$@" 2>&1 | grep -v amdgpu.ids > $LOG; cat $LOG ;;
  prof)       # rocprofv3 kernel stats of a short run of one model (single-stream so that durations are attributable): prof swin_s
    M=${1:-swin_s}; shift; TAG=${PTAG:-$M}      # extra arguments go to bench.py; PTAG names the output; SIDE=1 keeps the second stream
    cd /tmp
    mkdir -p $R/gpurun_out/r6_prof_$TAG
    VTX_SIDE_WGRAD=${SIDE:-0} timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6_prof_$TAG -o trace -- python $R/bench.py --model $M --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-kernel-events "$@" > $R/gpurun_out/r6_prof_$TAG.log 2>&1
    cd $R
    python tools/rocpd_stats.py gpurun_out/r6_prof_$TAG/trace_results.db --steps 7 --top 70 > gpurun_out/r6_kernel_stats_$TAG.md 2>> gpurun_out/r6_prof_$TAG.log || true
    [ -n "$OVERLAP" ] && python tools/r6/overlap_of.py gpurun_out/r6_prof_$TAG/trace_results.db "$OVERLAP" > gpurun_out/r6_overlap_$TAG.txt 2>&1
    [ -n "$GAPS" ] && python tools/rocpd_gaps.py gpurun_out/r6_prof_$TAG/trace_results.db --skip 0.5 > gpurun_out/r6_gaps_$TAG.txt 2>&1
    rm -f gpurun_out/r6_prof_$TAG/trace_results.db
    M=$TAG
    head -40 gpurun_out/r6_kernel_stats_$M.md ;;
  battery)    # end-of-round measurements: bench lines of the five workloads, PMC traffic, GEMM micro-benchmark vs hipBLASLt, shape table
    timeout 1200 python bench.py --steps 20 --warmup 5 --shape-table gpurun_out/r6_shape_table_swin_s.md 2>&1 | grep '"metric"' > gpurun_out/r6_bench_swin_s.json
    for m in vit_s16 pvt_small; do timeout 900 python bench.py --model $m --steps 20 --warmup 5 --shape-table gpurun_out/r6_shape_table_$m.md 2>&1 | grep '"metric"' > gpurun_out/r6_bench_$m.json; done
    timeout 900 python bench.py --model dino --steps 10 --warmup 3 --cpu-batch 2 --cpu-steps 1 2>&1 | grep '"metric"' > gpurun_out/r6_bench_dino.json
    timeout 900 python bench.py --model twins_svt_s --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '"metric"' > gpurun_out/r6_bench_twins_svt_s.json
    for m in swin_s vit_s16 pvt_small; do timeout 1500 tools/pmc_traffic.sh $m > /dev/null 2>&1; done
    timeout 900 python tools/bench_gemm.py --vendor 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_gemm_bench.log
    timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_attn_bench.log
    for m in swin_s vit_s16 pvt_small dino twins_svt_s; do cut -c1-200 gpurun_out/r6_bench_$m.json; done ;;
  py)         # any python tool: tools/r6/job.sh py name script.py args...
    N=$1; shift
    timeout ${TMO:-1200} python "$@" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_$N.log; tail -${TAIL:-60} gpurun_out/r6_$N.log ;;
  *) echo "unknown job $J"; exit 2 ;;
esac
