# A/B of env-switched variants at B = 256 and B = 1024: bash tools/run_ab.sh "VAR=a VAR=b ..."   (GPU box)
mkdir -p gpurun_out/ab
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=600 -k "golden or tox21_shape or dropout" > gpurun_out/ab/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/ab/pytest.log | tail -4
for v in $1; do
  for B in 256 1024; do
    env $v timeout 300 python bench.py --no-cpu-baseline --no-extras --batch $B --repeats 9 --steps 40 > gpurun_out/ab/b${B}_$v.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open('gpurun_out/ab/b${B}_$v.json').read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('$v B=$B', d['ms_per_step'], d['value_min'], d['value_max'], 'bn', k['bn'], 'head', k['head'], 'readout', k['readout'], 'agg', k['agg'])
PY
  done
done
