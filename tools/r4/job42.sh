#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job42; mkdir -p $O
for v in base pfd0 pfd4 pfd12 base pfd0; do
  if [ $v = base ]; then unset VTX_LIBVTX; else export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so; fi
  echo "== $v"; timeout 200 python tools/r4/wgrad_wide_check.py 2>&1 | grep "us ("
done | tee $O/pfd.log
