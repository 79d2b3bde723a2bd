#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job34; mkdir -p $O
for rep in 1 2; do for a in 1; do
  VTX_DGRAD_SPLITK=$a timeout 600 python bench.py --model dino --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' > $O/b.log
  echo "dino dgrad split-K=$a: $(python -c "import json;d=json.loads(open('$O/b.log').read());print(d['value'], d['ms_per_step'])")"
done; done
timeout 900 python -m pytest tests/test_gpu_dino.py -m gpu -x -q 2>&1 | tail -3
