#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job37; mkdir -p $O
for v in base wapf base wapf; do
  if [ $v = base ]; then unset VTX_LIBVTX; else export VTX_LIBVTX=$R/tools/probe/ablate/libvtx_$v.so; fi
  echo "== $v"; timeout 200 python tools/bench_attn.py 2>&1 | grep -v "amdgpu.ids\|^\[W" | head -12
done | tee $O/wattn.log
