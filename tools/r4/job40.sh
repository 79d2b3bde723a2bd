#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job40; mkdir -p $O
for v in base ww1 ww2 ww4 ww3; do
  if [ $v = base ]; then unset VTX_LIBVTX; else export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so; fi
  echo "== $v"; VTX_CHECK_SKIP=1 timeout 200 python tools/r4/wgrad_wide_check.py 2>&1 | grep "us ("
done | tee $O/ablate.log
