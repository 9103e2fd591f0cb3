# round 3, call C: molecule-staged backward (mol.hip) on/off, full GPU suite
mkdir -p gpurun_out/r3c
timeout 900 python -m pytest tests -m gpu -q --timeout=300 > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed|trajectory parity" gpurun_out/r3c/pytest.log | tail -12
EAGCN_MOLBWD=0 timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r3c/bench_mol0.json 2> gpurun_out/r3c/bench_mol0.err
EAGCN_MOLBWD=1 timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r3c/bench_mol1.json 2> gpurun_out/r3c/bench_mol1.err
python - <<'PY'
import json
for n in ('mol0','mol1'):
    try:
        d=json.loads(open('gpurun_out/r3c/bench_%s.json'%n).read().strip().splitlines()[-1])
        print(n, d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], d['kernel_ms_per_step'], {k:(v.get('ms_per_step'),v.get('step_frac')) for k,v in d.get('extra',{}).items()})
    except Exception as e:
        print(n, 'failed', e); print(open('gpurun_out/r3c/bench_%s.err'%n).read()[-1500:])
PY
