#!/bin/bash
# Same-box A/B of the current library against tools/probe/ablate/libvtx_base.so (tools/probe/build_base.sh), alternating:
#   tools/probe/ab_lib.sh [models...]   -> gpurun_out/ab_lib.log
R=$PWD; mkdir -p gpurun_out; L=gpurun_out/ab_lib.log; rm -f $L
MODELS=${@:-swin_s}
for rep in 1 2 3; do for m in $MODELS; do for which in base new; do
  if [ $which = base ]; then export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_base.so; else unset VTX_LIBVTX; fi
  echo -n "$m $which : " >> $L
  python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $L
done; done; done
cat $L
