mkdir -p gpurun_out
python -m pytest tests/test_gpu_dino.py tests/test_gpu_kernels.py tests/test_gpu_dispatch.py tests/test_gpu_models.py -q -m gpu -x --timeout 900 2>&1 | tail -6 | tee gpurun_out/t2.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '"metric"' > gpurun_out/bench_r3b.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r3b.json'))
print(d['value'], d['ms_per_step'])
for s in d['secondary']: print(s['model'], s['value'], s['ms_per_step'])
for k,v in d['roofline']['kernels'].items(): print(f"{k[:64]:64s} n={v['launches_per_step']:6} ms={v['ms_per_step']:7.3f} us={v['avg_launch_us']:7.2f}")
PY
