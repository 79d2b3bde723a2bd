#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job23; mkdir -p $O
for rep in 1 2; do
for m in swin_s vit_s16; do
  for v in ntsel ntall; do
    if [ $v = base ]; then unset VTX_LIBVTX; else export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so; fi
    timeout 600 python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' > $O/bench_${m}_$v.log
    echo "$m $v: $(python -c "import json;d=json.loads(open('$O/bench_${m}_$v.log').read());print(d['value'], d['ms_per_step'])")"
  done
done
done
