#!/bin/bash
# libvtx with the two-group GEMM's ablation variants compiled in (VTX_PP_ABLATE): tools/r5/ablate/libvtx_pp.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/vision-transformers-pytorch_amd/csrc
python -c "import sys; sys.path.insert(0, '$R/vision-transformers-pytorch_amd'); from vtx import build; build.build()"
mkdir -p $R/tools/r5/ablate
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value -DVTX_PP_ABLATE "$@" -c $C/gemm_pp.hip -o /tmp/gemm_pp_abl.o
objs=$(ls $C/build/*.o | grep -v gemm_pp.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gemm_pp_abl.o -o $R/tools/r5/ablate/libvtx_pp.so
ls -la $R/tools/r5/ablate/
