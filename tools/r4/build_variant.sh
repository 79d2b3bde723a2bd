#!/bin/bash
# Build the CURRENT tree with extra compiler flags into tools/probe/ablate/libvtx_<name>.so (selected at run time through VTX_LIBVTX):
#   tools/r4/build_variant.sh <name> "<extra flags>" [files...]      (files: only these .hip are recompiled with the flags; the others are
#   taken from csrc/build/*.o -- run `python -m vtx.build` / __graft_entry__.build() first)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; EXTRA=$2; shift 2
C=$R/vision-transformers-pytorch_amd/csrc
T=/tmp/vtx_var_$NAME; rm -rf $T; mkdir -p $T $R/tools/probe/ablate
objs=""
for f in $C/*.hip; do
  b=$(basename $f .hip)
  if [ $# -eq 0 ] || [[ " $* " == *" $b.hip "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value $EXTRA -c $f -o $T/$b.o &
    objs="$objs $T/$b.o"
  else
    objs="$objs $C/build/$b.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/tools/probe/ablate/libvtx_$NAME.so
ls -la $R/tools/probe/ablate/libvtx_$NAME.so
