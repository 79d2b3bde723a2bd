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
  strip)      # strip GEMM: bitwise vs tiled, race screen, timing vs tiled / hipBLASLt; vendor kernel names
    timeout 900 python tools/r5/strip_check.py "$@" 2>&1 | grep -v amdgpu.ids > $LOG
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5_vendor -o v -- python $R/tools/r5/vendor_names.py > $R/gpurun_out/r5_vendor.log 2>&1)
    python tools/rocpd_stats.py gpurun_out/r5_vendor/v_results.db --steps 1 --top 40 > gpurun_out/r5_vendor_kernels.md 2>&1
    rm -f gpurun_out/r5_vendor/v_results.db
    tail -40 $LOG ;;
  tests)      # the GPU test suite (optionally -k expr)
    timeout 2400 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -30 > $LOG; tail -30 $LOG ;;
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
  *) echo "unknown job $J"; exit 2 ;;
esac
