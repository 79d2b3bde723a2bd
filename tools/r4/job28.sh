#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job28; mkdir -p $O
for rep in 1 2; do for a in 1 3 2; do
  VTX_GEMM_ASTAT=$a timeout 600 python bench.py --model swin_s --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' > $O/b.log
  echo "swin_s astat=$a: $(python -c "import json;d=json.loads(open('$O/b.log').read());print(d['value'], d['ms_per_step'])")"
done; done
