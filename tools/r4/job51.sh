#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job51; mkdir -p $O
for v in base pref1 base pref1 base pref1; do
  if [ $v = base ]; then unset VTX_LIBVTX; else export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so; fi
  echo "== $v"; timeout 200 python tools/r4/wgrad_wide_check.py --case "s" 2>&1 | grep "us ("
done | tee $O/pref.log
