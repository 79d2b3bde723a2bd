mkdir -p gpurun_out; rm -f gpurun_out/ab_compact.log
for rep in 1 2; do for o in 0 1; do
  for m in swin_s vit_s16 pvt_small; do
  echo -n "model $m VTX_DP_COMPACT=$o : " | tee -a gpurun_out/ab_compact.log
  VTX_DP_COMPACT=$o python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a gpurun_out/ab_compact.log
  done
done; done
