# round 3, call D: molecule-staged backward with sharded tickets + fused BatchNorm finalizes; A/B by env
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests -m gpu -q --timeout=300 > gpurun_out/r3d/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed|trajectory parity" gpurun_out/r3d/pytest.log | tail -12
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  EAGCN_MOLBWD=$1 EAGCN_BN_FUSED=$2 timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r3d/bench_m$1f$2.json 2> gpurun_out/r3d/bench_m$1f$2.err
done
python - <<'PY'
import json
for n in ('m0f0','m1f0','m0f1','m1f1'):
    try:
        d=json.loads(open('gpurun_out/r3d/bench_%s.json'%n).read().strip().splitlines()[-1])
        print(n, d['value'], d['ms_per_step'], d['kernel_ms_per_step'], {k:(v.get('ms_per_step'),v.get('step_frac')) for k,v in d.get('extra',{}).items()})
    except Exception as e:
        print(n, 'failed', e); print(open('gpurun_out/r3d/bench_%s.err'%n).read()[-1500:])
PY
