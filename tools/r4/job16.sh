#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job16; mkdir -p $O
timeout 300 python tools/r4/astat_check.py --iters 20 2>&1 | grep -v "^\[W\|amdgpu.ids" | cut -c1-200 > $O/astat_swin.log; cat $O/astat_swin.log
timeout 300 python tools/r4/astat_check.py --vit --iters 20 2>&1 | grep -v "^\[W\|amdgpu.ids" | cut -c1-200 > $O/astat_vit.log; cat $O/astat_vit.log
export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_tr1.so
for k in bias resid; do
  N=1152; [ $k = resid ] && N=384
  timeout 300 python tools/r4/astat_trace.py --kind $k --N $N 2>&1 | grep -v "^\[W\|amdgpu.ids" > $O/trace_wg1_$k.log
done
for f in $O/trace*.log; do echo $f; tail -n 1 $f; done; sed -n 1,30p $O/trace_wg1_bias.log
