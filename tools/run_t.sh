mkdir -p gpurun_out/tests
timeout 800 python -m pytest tests -m gpu -q --timeout=300 > gpurun_out/tests/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/tests/pytest.log | tail -8
timeout 300 python bench.py --no-cpu-baseline --repeats 7 2>&1 | tail -1 > gpurun_out/bench_now.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_now.json'))
print(d['value'], d['ms_per_step'], d['roofline'].get('step_frac'), d['roofline'].get('frac'), d['roofline'].get('achieved'))
for k,v in d.get('extra',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('step_frac'))
PY
