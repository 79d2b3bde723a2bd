#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job10; mkdir -p $O
timeout 600 python tools/r4/astat_check.py 2>&1 | grep -v "^\[W\|amdgpu.ids" > $O/astat_swin.log; cat $O/astat_swin.log
timeout 600 python tools/r4/astat_check.py --vit 2>&1 | grep -v "^\[W\|amdgpu.ids" > $O/astat_vit.log; cat $O/astat_vit.log
