mkdir -p gpurun_out/r3k
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=600 -k "golden or tox21_shape or dropout or without_any_bond" > gpurun_out/r3k/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r3k/pytest.log | tail -5
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3k/bench.json 2> gpurun_out/r3k/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3k/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], {k:(v.get('ms_per_step'), v.get('step_frac'), v.get('relevant_frac')) for k,v in d.get('extra',{}).items()})
PY
bash tools/run_prof.sh r3k_b1024 --steps 10 --warmup 3 --batch 1024 > gpurun_out/r3k/prof_b1024.txt 2>&1
grep -E "agg|index_scan|bn_|readout|head|gemm|unpack|pack" gpurun_out/prof_r3k_b1024/summary.txt | cut -c1-60,112-190 | head -32
bash tools/run_prof.sh r3k_b256 --steps 20 --warmup 5 > gpurun_out/r3k/prof_b256.txt 2>&1
grep -E "agg|index_scan|bn_|readout|head|gemm|unpack|pack|loss" gpurun_out/prof_r3k_b256/summary.txt | cut -c1-60,112-190 | head -32
