#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job43; mkdir -p $O
for v in ww2 ww10 ww8 ww2 ww10; do
  export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so
  echo "== $v"; VTX_CHECK_SKIP=1 timeout 200 python tools/r4/wgrad_wide_check.py --wide 1 --case "s" 2>&1 | grep "us ("
done | tee $O/same_slice.log
