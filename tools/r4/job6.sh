#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job6; mkdir -p $O
# the fixed library, no fence (now the default): call-by-call (the path that showed 10-25 %) and the one-call path
VTX_LAYER_CALL=0 timeout 1200 python tools/probe/determinism_stress.py swin_s 2001 2>&1 | grep -v "^\[W" | tail -5 > $O/stress_callbycall.log; cat $O/stress_callbycall.log | cut -c1-300
timeout 900 python tools/probe/determinism_stress.py swin_s 801 2>&1 | grep -v "^\[W" | tail -3 > $O/stress_onecall.log; cat $O/stress_onecall.log | cut -c1-300
BISECT_MODE=light VTX_LAYER_CALL=0 timeout 900 python tools/probe/merge_bisect.py 300 2>&1 | grep "merge_bisect:" | cut -c1-300 > $O/bisect_fixed.log; cat $O/bisect_fixed.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '"metric"' | cut -c1-200
tools/gpu_check.sh tests
