#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job3; mkdir -p $O
timeout 300 tools/probe/dpp_pk_hazard.bin > $O/hazard.log 2>&1; cat $O/hazard.log
for v in gs1 gs2 default; do
  if [ $v = default ]; then unset VTX_LIBVTX; else export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so; fi
  BISECT_MODE=light VTX_LAYER_CALL=0 VTX_SIDE_FENCE=0 timeout 900 python tools/probe/merge_bisect.py 200 2>&1 | grep -v "^\[W" > $O/bisect_$v.log
  echo "== $v"; grep -E "merge_bisect:|first differing" $O/bisect_$v.log | tail -4 | cut -c1-300
done
