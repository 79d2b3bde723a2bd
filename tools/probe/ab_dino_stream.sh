#!/bin/bash
L=gpurun_out/ab_dino_stream.log; rm -f $L
for rep in 1 2 3; do for t in 0 1; do
  echo -n "dino VTX_DINO_TEACHER_STREAM=$t : " >> $L
  VTX_DINO_TEACHER_STREAM=$t python bench.py --model dino --steps 15 --warmup 4 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $L
done; done
cat $L
