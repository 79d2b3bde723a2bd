#!/bin/bash
# Round-end measurement battery (run through gpurun from the repo root); outputs under gpurun_out/.
TAG=${1:-r1}
tools/gpu_check.sh tests bench prof:$TAG profvit:$TAG
python bench.py --model vit_s16 --steps 20 --warmup 5 2>&1 | grep metric > gpurun_out/bench_vit.log
tools/pmc_traffic.sh swin_s > /dev/null 2>&1
tools/pmc_traffic.sh vit_s16 > /dev/null 2>&1
tools/pmc_traffic.sh pvt_small > /dev/null 2>&1
python tools/bench_gemm.py --vendor 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_bench.log
python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids > gpurun_out/attn_bench.log
