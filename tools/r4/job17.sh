#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job17; mkdir -p $O
for v in base; do
  if [ $v = base ]; then unset VTX_LIBVTX; else export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so; fi
  echo "== $v"; timeout 60 python tools/r4/astat_check.py --iters 20 2>&1 | grep -E "qkv fwd|fc1 fwd|fc2 dgrad|proj|K 192|equal: False|hang" | cut -c1-112
done > $O/ablate.log 2>&1
cat $O/ablate.log
