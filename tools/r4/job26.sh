#!/bin/bash
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/r4job26; mkdir -p $O
timeout 600 python tools/r4/layer_diff.py 2>&1 | grep -v "amdgpu.ids\|^\[W" | awk '$(NF-8)+0 > 0 || $NF+0 > 0 || /logits/' | head -20
timeout 1200 python -m pytest tests/test_gpu_dispatch.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python -m pytest "tests/test_gpu_models.py::test_full_size_bf16_step_is_deterministic_finite_and_matches_the_chunked_path" -x -q 2>&1 | tail -3
