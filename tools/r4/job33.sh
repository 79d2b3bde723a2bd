#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job33; mkdir -p $O
for v in base so0; do
  if [ $v = base ]; then unset VTX_LIBVTX; else export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so; fi
  echo "== $v"; timeout 120 python tools/r4/fc1_kinds.py 2>&1 | grep "^M "
  timeout 200 python tools/r4/astat_check.py --iters 20 2>&1 | grep -E "qkv fwd|proj|fc2 dgrad|K 192" | cut -c1-150
done | tee $O/split_order.log
