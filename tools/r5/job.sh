#!/bin/bash
# Round-5 GPU jobs, one parameterised script (run through gpurun from the repo root): tools/r5/job.sh <job> [args]
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'tools/r5/job.sh strip'
# Everything a job prints goes to gpurun_out/r5_<job>.log; profiles to gpurun_out/r5_<job>_*.
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
J=$1; shift
LOG=gpurun_out/r5_$J.log
case "$J" in
  pp)         # two-group GEMM: bitwise vs the other kernels, race screen, timing vs them and hipBLASLt; phase ablation
    timeout 1200 python tools/r5/pp_check.py "$@" 2>&1 | grep -v amdgpu.ids > $LOG
    [ -f tools/r5/ablate/libvtx_pp.so ] || bash tools/r5/build_pp_ablate.sh > /dev/null 2>&1       # (the ablation build: hipcc is on the GPU box too)
    VTX_LIBVTX=$R/tools/r5/ablate/libvtx_pp.so timeout 600 python tools/r5/pp_ablate.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_pp_ablate.log
    tail -50 $LOG; cat gpurun_out/r5_pp_ablate.log ;;
  skew)       # two-group GEMM: start delay per column tile (VTX_PP_SKEW, units of ~0.43 us)
    : > $LOG
    for k in "$@"; do echo "== VTX_PP_SKEW=$k" >> $LOG; VTX_PP_SKEW=$k timeout 600 python tools/r5/pp_check.py --quick --no-correctness 2>&1 | grep -v "amdgpu.ids\|^CUs" >> $LOG; done
    cat $LOG ;;
  tests)      # the GPU test suite (optionally -k expr)
    if [ $# -gt 0 ] && [ -e "$1" ]; then T="$1"; shift; else T=tests; fi       # first argument: a test file, else the whole suite
    timeout 2400 python -m pytest $T -m gpu -x -q "$@" 2>&1 | tail -30 > $LOG; tail -30 $LOG ;;
  bench)      # headline + secondaries
    timeout 900 python bench.py --steps 20 --warmup 5 "$@" 2>&1 | grep '"metric"' > $LOG; cut -c1-400 $LOG ;;
  ab)         # same-box A/B of one option on one model: tools/r5/job.sh ab swin_s GEMM_STRIP 0 1
    M=$1; O=$2; shift 2
    : > $LOG
    for rep in 1 2; do for v in "$@"; do
      echo "== $O=$v" >> $LOG
      env VTX_$O=$v timeout 600 python bench.py --model $M --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '"metric"' | cut -c1-200 >> $LOG
    done; done
    cat $LOG ;;
  ln)         # LayerNorm rows-in-flight variants (VTX_LN_ROWS bits): micro-benchmark + the LayerNorm tests under each
    : > $LOG
    for v in 0 1 2 3; do
      echo "== VTX_LN_ROWS=$v" >> $LOG
      VTX_LN_ROWS=$v timeout 300 python tools/bench_ln.py 2>&1 | grep -v amdgpu.ids >> $LOG
    done
    VTX_LN_ROWS=3 timeout 900 python -m pytest tests -m gpu -x -q -k "layernorm or ln_ or merge" 2>&1 | tail -3 >> $LOG
    cat $LOG ;;
  wfast)      # window-attention forward under WATTN_FAST: the option test, the micro-benchmark, the train step
    timeout 900 python -m pytest tests/test_gpu_dispatch.py -m gpu -x -q -k "wattn or window" 2>&1 | tail -15 > $LOG
    timeout 600 python tools/r5/wattn_fast_check.py 2>&1 | grep -v amdgpu.ids >> $LOG
    for rep in 1 2; do for v in "0 VTX_WATTN_FWD4=0" "3 VTX_WATTN_FWD4=0" "3 VTX_WATTN_FWD4=1"; do
      echo "== WATTN_FAST=$v" >> $LOG
      env VTX_WATTN_FAST=$v timeout 600 python bench.py --model swin_s --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '"metric"' | cut -c1-200 >> $LOG
    done; done
    cat $LOG ;;
  mlp)        # fused MLP of the narrow stages: parity tests, micro-benchmark vs the four GEMM launches, same-box A/B in Swin-S
    timeout 1200 python -m pytest tests/test_gpu_mlp_fused.py -m gpu -q 2>&1 | tail -40 > $LOG
    timeout 600 python tools/r5/mlp_fused_bench.py 2>&1 | grep -v amdgpu.ids >> $LOG
    for rep in 1 2; do for v in ${MLPV:-0 1}; do
      echo "== MLP_FUSED=$v" >> $LOG
      env VTX_MLP_FUSED=$v timeout 600 python bench.py --model ${MLPM:-swin_s} --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '"metric"' | cut -c1-200 >> $LOG
    done; done
    cat $LOG ;;
  *) echo "unknown job $J"; exit 2 ;;
esac
