#!/bin/bash
# libvtx with phases of the window-attention kernels compiled out (WA_ABLATE in csrc/attention_win.hip): tools/r5/ablate/libvtx_wa<bits>.so
#   bash tools/r5/build_wattn_ablate.sh 32 48
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/vision-transformers-pytorch_amd/csrc
mkdir -p $R/tools/r5/ablate
objs=$(ls $C/build/*.o | grep -v "/attention_win.o")
for n in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value -DWA_ABLATE=$n -c $C/attention_win.hip -o /tmp/wa_abl_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/wa_abl_$n.o -o $R/tools/r5/ablate/libvtx_wa$n.so ) &
done
wait
ls -la $R/tools/r5/ablate/
