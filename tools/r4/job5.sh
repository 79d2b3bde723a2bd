#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job5; mkdir -p $O
for v in gs6 gs7 default; do
  if [ $v = default ]; then unset VTX_LIBVTX; else export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so; fi
  BISECT_MODE=light VTX_LAYER_CALL=0 VTX_SIDE_FENCE=0 timeout 900 python tools/probe/merge_bisect.py 300 2>&1 | grep -v "^\[W" > $O/bisect_$v.log
  echo "== $v"; grep -E "merge_bisect:" $O/bisect_$v.log | tail -1 | cut -c1-300
done
