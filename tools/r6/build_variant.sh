#!/bin/bash
# A second libvtx.so with ONE source file compiled under extra defines (same-box A/B through VTX_LIBVTX: tools/r6/job.sh ablib):
#   tools/r6/build_variant.sh gemm_astat.hip astat_nt3 -DASTAT_NT=3     -> tools/r6/variants/libvtx_astat_nt3.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/vision-transformers-pytorch_amd/csrc
SRC=$1; NAME=$2; shift 2
python -c "import sys; sys.path.insert(0, '$R/vision-transformers-pytorch_amd'); from vtx import build; build.build(verbose=False)"
mkdir -p $R/tools/r6/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value "$@" -c $C/$SRC -o /tmp/variant_$NAME.o
objs=$(ls $C/build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/variant_$NAME.o -o $R/tools/r6/variants/libvtx_$NAME.so
echo built tools/r6/variants/libvtx_$NAME.so
