# round 3, call A: XCD-local GEMM schedule on/off (micro + step), full GPU suite
mkdir -p gpurun_out/r3a
for t in 4809 19200; do
  T=$t timeout 300 python tools/gemm_sk_bench.py > gpurun_out/r3a/gemm_T$t.txt 2>&1
done
EAGCN_GEMM3_XK=0 timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r3a/bench_xk0.json 2> gpurun_out/r3a/bench_xk0.err
EAGCN_GEMM3_XK=1 timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r3a/bench_xk1.json 2> gpurun_out/r3a/bench_xk1.err
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -x > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r3a/pytest.log | tail -6
cat gpurun_out/r3a/gemm_T*.txt | tr '|' '\n' | grep -E "pair|T=" 
python - <<'PY'
import json
for n in ('xk0','xk1'):
    try:
        d=json.loads(open('gpurun_out/r3a/bench_%s.json'%n).read().strip().splitlines()[-1])
        print(n, d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg_us', d['roofline']['avg_launch_us'], {k:(v.get('ms_per_step'),v.get('step_frac')) for k,v in d.get('extra',{}).items()})
    except Exception as e:
        print(n, 'failed', e)
PY
