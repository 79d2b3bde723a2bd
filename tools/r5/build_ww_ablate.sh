#!/bin/bash
# libvtx with a timing-only variant of the wide-tile weight gradient (WW_ABLATE bits: 1 no DMA requests | 2 no fragment reads + MFMA |
# 4 no slab / output stores | 8 every slice streams slice 0's tokens): tools/r5/ablate/libvtx_ww<bits>.so -- results are garbage
#   tools/r5/build_ww_ablate.sh 4
set -e
B=${1:-4}
shift || true
X=$(echo "$@" | tr -cd "A-Za-z0-9=_")
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/vision-transformers-pytorch_amd/csrc
python -c "import sys; sys.path.insert(0, '$R/vision-transformers-pytorch_amd'); from vtx import build; build.build(verbose=False)"
mkdir -p $R/tools/r5/ablate
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value -DWW_ABLATE=$B "$@" -c $C/gemm_wgrad_glds.hip -o /tmp/gemm_wgrad_abl$B.o
objs=$(ls $C/build/*.o | grep -v gemm_wgrad_glds.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gemm_wgrad_abl$B.o -o $R/tools/r5/ablate/libvtx_ww$B$X.so
