#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job4; mkdir -p $O
timeout 300 tools/probe/dpp_pk_hazard.bin > $O/hazard.log 2>&1; cat $O/hazard.log
for v in gs3 gs4 gs1; do
  export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so
  BISECT_MODE=light VTX_LAYER_CALL=0 VTX_SIDE_FENCE=0 timeout 900 python tools/probe/merge_bisect.py 300 2>&1 | grep -v "^\[W" > $O/bisect_$v.log
  echo "== $v"; grep -E "merge_bisect:" $O/bisect_$v.log | tail -1 | cut -c1-300
done
