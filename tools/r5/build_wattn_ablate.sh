#!/bin/bash
# libvtx with variants of the window-attention kernels (csrc/attention_win.hip): tools/r5/ablate/libvtx_<name>.so
#   bash tools/r5/build_wattn_ablate.sh wa32:-DWA_ABLATE=32 wa48:-DWA_ABLATE=48 occ6:-DWF4_OCC=6
# WA_ABLATE: phases compiled out (32: forward loads, LDS staging and stores only; 16: no bias-table build); WF4_OCC: workgroups per CU of the four-wave forward
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/vision-transformers-pytorch_amd/csrc
mkdir -p $R/tools/r5/ablate
objs=$(ls $C/build/*.o | grep -v "/attention_win.o")
for nv in "$@"; do
  n=${nv%%:*}; f=${nv#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value ${f//,/ } -c $C/attention_win.hip -o /tmp/wa_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/wa_$n.o -o $R/tools/r5/ablate/libvtx_$n.so ) &
done
wait
ls -la $R/tools/r5/ablate/
