#!/bin/bash
# round 4, job 1: baseline bench on this box, determinism bisect, SQ counters of the LDS-DMA GEMMs in the model, fill/copy neighbours
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job1; mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '"metric"' > $O/bench_base.json
cut -c1-200 $O/bench_base.json
VTX_LAYER_CALL=0 VTX_SIDE_FENCE=0 timeout 600 python tools/probe/merge_bisect.py 60 2>&1 | grep -v "^\[W" > $O/bisect_callbycall.log
tail -30 $O/bisect_callbycall.log
VTX_SIDE_WGRAD=0 tools/pmc_kernel.sh r4gemm gemm_glds -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary > $O/pmc_gemm.txt 2>&1
tools/gpu_check.sh prof:r4a > $O/prof.log 2>&1
tools/probe/prof_bygrid.sh swin_s > /dev/null 2>&1
echo done
