mkdir -p gpurun_out
python -m pytest tests/test_gpu_dispatch.py -q -m gpu --timeout 900 -k "one_call" 2>&1 | tail -3
python tools/probe/host_time.py 2>&1 | grep -v amdgpu | tee gpurun_out/host_time_r3b.log
VTX_LAYER_CALL=0 python tools/probe/host_time.py 2>&1 | grep -v amdgpu | sed 's/^/[call-by-call] /' | tee -a gpurun_out/host_time_r3b.log
