#!/bin/bash
# debug builds of csrc/attention.hip (the fp32 D = 32, 10-key-tile instantiation): tools/r5/ablate/libvtx_attn_<v>.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/vision-transformers-pytorch_amd/csrc
mkdir -p $R/tools/r5/ablate
F="--offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value"
objs=$(ls $C/build/*.o | grep -v "/attention.o")
build() { /opt/rocm/bin/hipcc $F $2 -c $C/attention.hip -o /tmp/attn_$1.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/attn_$1.o -o $R/tools/r5/ablate/libvtx_attn_$1.so; }
for n in "$@"; do build nop$n "-O3 -DVTX_DBG_MFMA_NOP=$n" & done
wait
ls -la $R/tools/r5/ablate/
