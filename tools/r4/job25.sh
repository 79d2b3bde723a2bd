#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job25; mkdir -p $O
for cfg in "VTX_GEMM_ASTAT=1" "VTX_GEMM_ASTAT=1 VTX_SIDE_WGRAD=0" "VTX_GEMM_ASTAT=0"; do
  echo "== $cfg"; env $cfg timeout 300 python tools/r4/det_check.py --trials 4 2>&1 | grep "trial\|differ"
  echo "== $cfg fwd-only"; env $cfg timeout 300 python tools/r4/det_check.py --trials 4 --fwd-only 2>&1 | grep "trial\|differ"
done > $O/det.log 2>&1
cat $O/det.log
