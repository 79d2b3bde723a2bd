#!/bin/bash
# Round-end measurement battery (run through gpurun from the repo root); outputs under gpurun_out/.
TAG=${1:-r1}
tools/gpu_check.sh tests bench prof:$TAG profvit:$TAG profpvt:$TAG
for m in vit_s16 pvt_small; do python bench.py --model $m --steps 20 --warmup 5 2>&1 | grep metric > gpurun_out/bench_$m.log; done
python bench.py --model dino --steps 10 --warmup 3 --cpu-batch 2 --cpu-steps 1 2>&1 | grep metric > gpurun_out/bench_dino.log
for m in swin_s vit_s16 pvt_small; do tools/pmc_traffic.sh $m > /dev/null 2>&1; done
python tools/bench_gemm.py --vendor 2>&1 | grep -v amdgpu.ids > gpurun_out/gemm_bench.log
python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids > gpurun_out/attn_bench.log
python tools/bench_input.py 2>&1 | grep -v amdgpu.ids > gpurun_out/input_bench.log
