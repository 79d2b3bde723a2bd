#!/bin/bash
# repeat the full-size Swin-S determinism test N times per setting, print failures
N=${1:-25}
for v in 1; do
  f=0
  for i in $(seq $N); do
    VTX_WATTN_BWD4=$v timeout 300 python -m pytest "tests/test_gpu_models.py::test_full_size_bf16_step_is_deterministic_finite_and_matches_the_chunked_path[swin_s]" -q -m gpu -x 2>&1 | grep -v "^\[W" > /tmp/st.log
    if ! grep -q "1 passed" /tmp/st.log; then f=$((f+1)); echo "--- VTX_WATTN_BWD4=$v run $i failed"; grep -E "assert|Error|error|mismatch|differ" /tmp/st.log | head -8; fi
  done
  echo "VTX_WATTN_BWD4=$v: $f failures of $N"
done
