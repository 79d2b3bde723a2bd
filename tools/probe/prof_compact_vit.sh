R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
for o in 0 1; do
  mkdir -p gpurun_out/profv_c$o
  (cd /tmp && VTX_DP_COMPACT=$o VTX_SIDE_WGRAD=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profv_c$o -o trace -- \
     python $R/bench.py --model vit_s16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-secondary > $R/gpurun_out/profv_c$o/run.log 2>&1)
  grep '"metric"' gpurun_out/profv_c$o/run.log | cut -c1-120
  python tools/rocpd_stats.py gpurun_out/profv_c$o/trace_results.db --steps 7 --top 40 > gpurun_out/profv_c$o/kernel_stats.md
  rm -f gpurun_out/profv_c$o/trace_results.db
done
