for cfg in "4809 400 704" "2500 304 512"; do
    set -- $cfg
    T=$1 FIN=$2 FP=$3 timeout 100 python tools/gemm_sk_bench.py 2>&1 | grep -v amdgpu.ids | tail -1
done
mkdir -p gpurun_out/tests
timeout 800 python -m pytest tests -m gpu -q --timeout=300 > gpurun_out/tests/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/tests/pytest.log | tail -8
timeout 300 python bench.py --no-cpu-baseline --repeats 7 2>&1 | tail -1 > gpurun_out/bench_now.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_now.json'))
print(d['value'], d['ms_per_step'], d['roofline'].get('step_frac'), d['roofline'].get('frac'))
for k,v in d.get('extra',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('step_frac'))
PY
