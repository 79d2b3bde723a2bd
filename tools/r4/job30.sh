#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job30; mkdir -p $O
for v in base lnrows2 base lnrows2; do
  if [ $v = base ]; then unset VTX_LIBVTX; else export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so; fi
  echo "== $v"; timeout 120 python tools/bench_ln.py 2>&1 | grep rows=
done | tee $O/ln.log
