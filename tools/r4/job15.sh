#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job15; mkdir -p $O
for w in 1 101; do
  export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_tr$w.so
  for k in bias resid; do
    N=1152; [ $k = resid ] && N=384
    timeout 300 python tools/r4/astat_trace.py --kind $k --N $N 2>&1 | grep -v "^\[W\|amdgpu.ids" > $O/trace_wg${w}_$k.log
  done
done
for f in $O/*.log; do echo $f; tail -n 1 $f; done; sed -n 1,60p $O/trace_wg1_bias.log
