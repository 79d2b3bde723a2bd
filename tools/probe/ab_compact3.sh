mkdir -p gpurun_out; rm -f gpurun_out/ab_compact3.log
for rep in 1 2; do for cfg in "0 0" "1 14" "1 18" "1 22"; do
  set -- $cfg
  for m in swin_s vit_s16; do
  echo -n "model $m COMPACT=$1 MIN=$2 : " | tee -a gpurun_out/ab_compact3.log
  VTX_DP_COMPACT=$1 VTX_DP_COMPACT_MIN=$2 python bench.py --model $m --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-events 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a gpurun_out/ab_compact3.log
  done
done; done
