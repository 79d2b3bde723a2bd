#!/bin/bash
# Round-end measurement battery (run through gpurun from the repo root); outputs under gpurun_out/.
#   tools/battery.sh prof TAG    tests + rocprofv3 kernel stats (4 workloads) + PMC traffic (3 models) + micro-benchmarks
#   tools/battery.sh bench       the bench.py lines of the 4 workloads (run AFTER the PMC json files were copied to profiles/)
R=$PWD
export TMPDIR=/tmp
case "$1" in
  prof)
    TAG=${2:-r1}
    tools/gpu_check.sh tests prof:$TAG profvit:$TAG profpvt:$TAG
    mkdir -p gpurun_out/profdino_$TAG
    (cd /tmp && VTX_SIDE_WGRAD=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/profdino_$TAG -o trace -- \
       python $R/bench.py --model dino --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/profdino_$TAG/run.log 2>&1)
    python tools/rocpd_stats.py gpurun_out/profdino_$TAG/trace_results.db --steps 5 --top 70 > gpurun_out/profdino_$TAG/kernel_stats.md
    rm -f gpurun_out/profdino_$TAG/trace_results.db
    for m in swin_s vit_s16 pvt_small; do timeout 1500 tools/pmc_traffic.sh $m > /dev/null 2>&1; done
    timeout 600 python tools/bench_gemm.py --vendor 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_bench.log
    timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids > gpurun_out/attn_bench.log
    timeout 300 python tools/bench_input.py 2>&1 | grep -v amdgpu.ids > gpurun_out/input_bench.log
    timeout 120 python tools/probe/hbm_floor.py 2>&1 | grep MB > gpurun_out/hbm_floor.log
    [ -x tools/probe/store_pattern.bin ] || hipcc --offload-arch=gfx950 -O3 -o tools/probe/store_pattern.bin tools/probe/store_pattern.hip > /dev/null 2>&1
    timeout 120 tools/probe/store_pattern.bin > gpurun_out/store_pattern.log 2>&1 ;;
  bench)
    timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep '"metric"' > gpurun_out/bench_swin_s.log
    for m in vit_s16 pvt_small; do timeout 900 python bench.py --model $m --steps 20 --warmup 5 2>&1 | grep '"metric"' > gpurun_out/bench_$m.log; done
    timeout 900 python bench.py --model dino --steps 10 --warmup 3 --cpu-batch 2 --cpu-steps 1 2>&1 | grep '"metric"' > gpurun_out/bench_dino.log
    timeout 900 python bench.py --model twins_svt_s --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '"metric"' > gpurun_out/bench_twins_svt_s.log
    for m in swin_s vit_s16 pvt_small dino twins_svt_s; do cut -c1-260 gpurun_out/bench_$m.log; done ;;
esac
