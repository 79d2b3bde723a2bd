#!/bin/bash
# per-shape (launch grid) durations of the hot kernels inside Swin-S: mapped (compacted) vs unmapped launches side by side
R=$PWD; export TMPDIR=/tmp
M=${1:-swin_s}
O=$R/gpurun_out/bygrid_$M; mkdir -p $O
(cd /tmp && VTX_SIDE_WGRAD=0 timeout 600 rocprofv3 --kernel-trace -d $O -o trace -- \
   python $R/bench.py --model $M --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-secondary > $O/run.log 2>&1)
for k in gemm_glds_pv wgrad_glds wattn_fwd wattn_bwd ln_fwd ln_bwd sattn; do
  python $R/tools/rocpd_stats.py $O/trace_results.db --steps 7 --by-grid $k
done > $O/bygrid.md
rm -f $O/trace_results.db
