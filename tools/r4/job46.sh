#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job46; mkdir -p $O
for m in swin_s vit_s16; do
  for rep in 1 2; do
  for cfg in "1 1" "0 1" "1 0"; do
    set -- $cfg
    echo -n "$m side=$1 wide=$2: "
    VTX_SIDE_WGRAD=$1 VTX_WGRAD_WIDE=$2 timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
  done
done | tee $O/ab.log
