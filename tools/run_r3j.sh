mkdir -p gpurun_out/r3j
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/r3j/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r3j/pytest.log | tail -8
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3j/bench.json 2> gpurun_out/r3j/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3j/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], {k:(v.get('ms_per_step'), v.get('step_frac'), v.get('relevant_frac')) for k,v in d.get('extra',{}).items()})
PY
