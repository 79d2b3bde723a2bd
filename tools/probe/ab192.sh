mkdir -p gpurun_out
python -m pytest tests/test_gpu_dispatch.py -q -m gpu -x --timeout 900 -k "glds" 2>&1 | tail -5 | tee gpurun_out/t192.log
for o in 0 1 2; do
  echo "== GLDS_BN192=$o stages 2,3" | tee -a gpurun_out/mb192.log
  VTX_GLDS_BN192=$o python tools/bench_gemm.py --stages 2,3 --what fwd,dgrad 2>&1 | grep -v amdgpu | tee -a gpurun_out/mb192.log
  echo "== GLDS_BN192=$o vit" | tee -a gpurun_out/mb192.log
  VTX_GLDS_BN192=$o python tools/bench_gemm.py --vit --what fwd,dgrad 2>&1 | grep -v amdgpu | tee -a gpurun_out/mb192.log
done
for rep in 1 2; do for o in 0 1 2; do
  for m in swin_s vit_s16; do
  echo "== model $m GLDS_BN192=$o" | tee -a gpurun_out/ab192.log
  VTX_GLDS_BN192=$o python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a gpurun_out/ab192.log
  done
done; done
