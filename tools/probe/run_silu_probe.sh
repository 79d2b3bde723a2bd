mkdir -p gpurun_out
bash tools/probe/build_ablate.sh 16 > /dev/null 2>&1
python tools/probe/silu_operand.py 2>&1 | grep -v amdgpu | tee gpurun_out/silu_probe.log
VTX_LIBVTX=$PWD/tools/probe/ablate/libvtx_a16.so python tools/probe/silu_operand.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/silu_probe.log
