#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job36; mkdir -p $O
run() { timeout 600 python bench.py --model $1 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' > $O/b.log; python -c "import json;d=json.loads(open('$O/b.log').read());print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  echo "swin_s default: $(run swin_s)"
  echo "swin_s astat=0: $(VTX_GEMM_ASTAT=0 run swin_s)"
done
timeout 300 python tools/r4/n384_check.py 2>&1 | grep "^M " | tee $O/stream.log
