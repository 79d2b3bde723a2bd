#!/bin/bash
# Libraries with one phase of the LDS-DMA GEMM removed (see GLDS_ABLATE in csrc/gemm_common.h), for in-model timing:
#   tools/probe/build_ablate.sh 1 4 5 8   ->  tools/probe/ablate/libvtx_a<N>.so ;  VTX_LIBVTX=<that> python bench.py ...
#   tools/probe/build_ablate.sh wgrad 1 2 4 8  ->  the same for the weight-gradient kernel (WG_ABLATE): libvtx_g<N>.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/vision-transformers-pytorch_amd/csrc
python -c "import sys; sys.path.insert(0, '$R/vision-transformers-pytorch_amd'); from vtx import build; build.build()"
mkdir -p $R/tools/probe/ablate
if [ "$1" = wgrad ]; then
  shift
  for n in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value -DWG_ABLATE=$n -c $C/gemm_wgrad_glds.hip -o /tmp/gemm_wgrad_glds_g$n.o
    objs=$(ls $C/build/*.o | grep -v gemm_wgrad_glds.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gemm_wgrad_glds_g$n.o -o $R/tools/probe/ablate/libvtx_g$n.so
  done
  ls -la $R/tools/probe/ablate/; exit 0
fi
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value -DGLDS_ABLATE=$n -c $C/gemm_glds.hip -o /tmp/gemm_glds_a$n.o
  objs=$(ls $C/build/*.o | grep -v gemm_glds.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gemm_glds_a$n.o -o $R/tools/probe/ablate/libvtx_a$n.so
done
ls -la $R/tools/probe/ablate/
