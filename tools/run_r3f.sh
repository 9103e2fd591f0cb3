# round 3, call F: multi-GPU plumbing on one rank (in-graph all-reduce, sync-BN hooks), full suite, default bench line
mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed|trajectory parity" gpurun_out/r3f/pytest.log | tail -12
EAGCN_FORCE_DIST=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 timeout 300 python tests/dist_multi_check.py > gpurun_out/r3f/dist_multi.log 2>&1; tail -4 gpurun_out/r3f/dist_multi.log
timeout 600 python bench.py > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3f/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['hbm']['step_frac'], d['roofline']['relevant'], d['roofline']['relevant_step_frac'])
print(d['kernel_ms_per_step'])
for k,v in d.get('extra',{}).items(): print(' ', k, v.get('value'), v.get('ms_per_step'), v.get('step_frac'), v.get('hbm_frac'), v.get('relevant_roofline'))
print(' cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('cores'), '| C1', d.get('cpu_baseline_configs0', {}).get('value'), d.get('cpu_baseline_configs0', {}).get('cores'))
PY
tail -3 gpurun_out/r3f/bench.err
