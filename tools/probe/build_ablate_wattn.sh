#!/bin/bash
# Libraries with one phase of the window-attention backward removed (WA_ABLATE in csrc/attention_win.hip):
#   tools/probe/build_ablate_wattn.sh 1 3 4 8 12  ->  tools/probe/ablate/libvtx_w<N>.so ;  VTX_LIBVTX=... python tools/bench_attn.py
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/vision-transformers-pytorch_amd/csrc
python -c "import sys; sys.path.insert(0, '$R/vision-transformers-pytorch_amd'); from vtx import build; build.build()"
mkdir -p $R/tools/probe/ablate
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-value -DWA_ABLATE=$n -c $C/attention_win.hip -o /tmp/attention_win_w$n.o
  objs=$(ls $C/build/*.o | grep -v attention_win.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/attention_win_w$n.o -o $R/tools/probe/ablate/libvtx_w$n.so
done
