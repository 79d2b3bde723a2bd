#!/bin/bash
# Build the library of an EARLIER commit next to the current one, for same-box A/B runs:
#   tools/probe/build_base.sh <git-rev>   ->  tools/probe/ablate/libvtx_base.so   (then: tools/probe/ab_lib.sh on the GPU box)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
REV=${1:-HEAD}
T=/tmp/vtx_base_src; rm -rf $T; mkdir -p $T $R/tools/probe/ablate
(cd $R && git archive $REV vision-transformers-pytorch_amd/csrc include | tar -x -C $T)
C=$T/vision-transformers-pytorch_amd/csrc
objs=""
for f in $C/*.hip; do
  o=$T/$(basename $f .hip).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value -I$T/include -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/tools/probe/ablate/libvtx_base.so
ls -la $R/tools/probe/ablate/libvtx_base.so
