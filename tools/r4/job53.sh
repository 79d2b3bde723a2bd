#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job53; mkdir -p $O
for v in base dsid base dsid; do
  if [ $v = base ]; then unset VTX_LIBVTX; else export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so; fi
  echo "== $v"; timeout 200 python tools/r4/astat_check.py 2>&1 | grep "dsilu\|fc1 fwd"; timeout 200 python tools/r4/astat_check.py --vit 2>&1 | grep "dsilu\|fc1 fwd"
done | tee $O/dsid.log
