#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job44; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dispatch.py -q -x -k "wgrad or compaction or one_call" 2>&1 | tail -5 | tee $O/tests.log
for m in swin_s vit_s16; do
  for w in 0 1 0 1; do
    echo "== $m wide=$w"
    VTX_WGRAD_WIDE=$w timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done | tee $O/ab.log
