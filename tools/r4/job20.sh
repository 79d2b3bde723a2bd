#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job20; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_dispatch.py -m gpu -x -q 2>&1 | tail -12 > $O/pytest_dispatch.log; cat $O/pytest_dispatch.log
for m in swin_s vit_s16; do
  for a in 0 1; do
    VTX_GEMM_ASTAT=$a timeout 600 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '"metric"' > $O/bench_${m}_astat$a.log
    echo "$m astat=$a: $(cut -c100-200 $O/bench_${m}_astat$a.log)"
  done
done
