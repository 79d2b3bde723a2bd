#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job18; mkdir -p $O
for v in tr38 tr0; do
  export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so
  timeout 120 python tools/r4/astat_trace.py --kind bias --N 1152 2>&1 | grep -v "^\[W\|amdgpu.ids" > $O/trace_$v.log
  echo "== $v"; sed -n 1,14p $O/trace_$v.log; tail -n 4 $O/trace_$v.log
done
