mkdir -p gpurun_out/r2g
python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/r2g/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r2g/pytest.log | tail -15
python bench.py --steps 20 --warmup 5 > gpurun_out/r2g/bench_default.json 2> gpurun_out/r2g/bench_default.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('gpurun_out/r2g/bench_default.json'))
print(d['value'],d['value_min'],d['value_max'],d['ms_per_step'],d['roofline']['achieved'],d['roofline']['frac'],d['roofline']['step_frac'])
print(d['kernel_ms_per_step']); print(json.dumps(d.get('extra'),indent=0)); print(d.get('cpu_baseline',{}).get('value'))"
tail -3 gpurun_out/r2g/bench_default.err
