mkdir -p gpurun_out/check
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/check/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/check/pytest.log | tail -8
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/check/bench.json 2> gpurun_out/check/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/check/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], {k:(v.get('ms_per_step'), v.get('step_frac'), v.get('relevant_frac')) for k,v in d.get('extra',{}).items()})
PY
