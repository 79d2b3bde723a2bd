#!/bin/bash
# same-box A/B of the A-stationary GEMM inside the models + the GPU tests that touch GEMMs
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job19; mkdir -p $O
for m in swin_s vit_s16; do
  for a in 0 1; do
    VTX_GEMM_ASTAT=$a timeout 600 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | grep '"metric"' | cut -c1-200 > $O/bench_${m}_astat$a.log
    echo "$m astat=$a: $(cat $O/bench_${m}_astat$a.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.log; cat $O/pytest.log
