#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job49; mkdir -p $O
for v in base prio1 prio2 base prio1 prio2; do
  if [ $v = base ]; then unset VTX_LIBVTX; else export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so; fi
  echo "== $v"; timeout 200 python tools/r4/wgrad_wide_check.py --wide 1 --case "s" 2>&1 | grep "us ("
done | tee $O/prio.log
