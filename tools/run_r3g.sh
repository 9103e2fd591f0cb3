# round 3, call G: streaming index scan, three-way gradient parity, GAT training vs RefGAT; bench
mkdir -p gpurun_out/r3g
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -s > gpurun_out/r3g/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed|trajectory parity|forward error vs" gpurun_out/r3g/pytest.log | tail -20
grep -E "e_hip .* e_ref" gpurun_out/r3g/pytest.log | sort -t' ' -k1,1 | head -60
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3g/bench.json 2> gpurun_out/r3g/bench.err; echo "bench rc=$?"
EAGCN_SCAN4=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3g/bench_scan1.json 2> gpurun_out/r3g/bench_scan1.err
python - <<'PY'
import json
for n in ('bench','bench_scan1'):
    d=json.loads(open('gpurun_out/r3g/%s.json'%n).read().strip().splitlines()[-1])
    print(n, d['value'], d['ms_per_step'], 'index', d['kernel_ms_per_step']['index'], {k:(v.get('ms_per_step')) for k,v in d.get('extra',{}).items()})
PY
