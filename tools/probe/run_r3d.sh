mkdir -p gpurun_out
rm -f gpurun_out/parity.log
timeout 1500 python -m pytest tests -q -m gpu -x --timeout 900 2>&1 | tail -5 | tee gpurun_out/tests_r3d.log
python tools/probe/host_time.py 2>&1 | grep -v amdgpu | tee gpurun_out/host_time_r3c.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '"metric"' > gpurun_out/bench_r3d.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r3d.json'))
print(d['value'], d['ms_per_step'])
for s in d['secondary']: print(s['model'], s['value'], s['ms_per_step'])
PY
