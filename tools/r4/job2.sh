#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job2; mkdir -p $O
VTX_LAYER_CALL=0 VTX_SIDE_FENCE=0 timeout 900 python tools/probe/merge_bisect.py 150 2>&1 | grep -v "^\[W" > $O/bisect2.log
tail -60 $O/bisect2.log
