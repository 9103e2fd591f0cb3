# Round-end verification on a GPU box: full GPU test suite, smoke, the default bench line, the profile set.
mkdir -p gpurun_out/tests gpurun_out/final
timeout 900 python -m pytest tests -m gpu -q --timeout=300 > gpurun_out/tests/pytest_final.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/tests/pytest_final.log | tail -6
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'step_frac', d['roofline']['step_frac'], 'traffic', d['roofline']['traffic'])
for k,v in d.get('extra',{}).items(): print(' ', k, v.get('value'), v.get('ms_per_step'), v.get('step_frac'))
print(' cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'))
PY
timeout 700 bash tools/run_r2_profiles.sh > gpurun_out/final/profiles.log 2>&1; echo "profiles rc=$?"
tail -3 gpurun_out/final/profiles.log
