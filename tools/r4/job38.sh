#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job38; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_twins.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do for a in 0 1; do
  VTX_TWINS_SUB_LDS=$a timeout 600 python bench.py --model twins_svt_s --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' > $O/b.log
  echo "twins_svt_s sub_lds=$a: $(python -c "import json;d=json.loads(open('$O/b.log').read());print(d['value'], d['ms_per_step'])")"
done; done
