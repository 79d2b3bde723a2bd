#!/bin/bash
# phase summary of the GEMM variants on the stage-3 / ViT shapes: tools/probe/gemm_trace_sweep.sh "128 642" "128 1" ...
cfgs=("$@"); [ ${#cfgs[@]} -eq 0 ] && cfgs=("128 642" "128 1" "64 642" "64 1")
for shape in "25088 1536 384 1 0" "25088 1152 384 0 0" "25088 384 1536 0 1" "25088 1536 384 2 0" "50432 1536 384 1 0" "50432 384 1536 0 1" "50432 384 384 0 1"; do
  for c in "${cfgs[@]}"; do
    timeout 60 tools/probe/gemm_trace.bin $shape $c 2>&1 | grep -E "per launch|^cold|main loop|prologue|stage pass|activation|mean wave" | sed 's/(with stamps), //; s/(operands evicted by a 640 MB memset before every launch)//; s/of wave lifetime//; s/mean //' | tr '\n' ' ' | sed 's/  */ /g'; echo
  done
done
