mkdir -p gpurun_out/r3i
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q --timeout=600 -k "fullsize or baseline_widths or tox21_shape or full_size" > gpurun_out/r3i/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r3i/pytest.log | tail -5
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3i/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:(v.get('ms_per_step'), v.get('step_frac'), v.get('relevant_frac')) for k,v in d.get('extra',{}).items()})
PY
