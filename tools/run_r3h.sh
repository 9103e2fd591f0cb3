# round 3, call H: float4 operand layout in the wave-per-tile aggregation (B > 512), parity slack calibration
mkdir -p gpurun_out/r3h
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/r3h/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r3h/pytest.log | tail -8
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3h/bench.json 2> gpurun_out/r3h/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3h/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], {k:(v.get('ms_per_step'), v.get('step_frac'), v.get('relevant_frac')) for k,v in d.get('extra',{}).items()})
PY
bash tools/run_prof.sh r3h_b1024 --steps 10 --warmup 3 --batch 1024 > gpurun_out/r3h/prof_b1024.txt 2>&1
grep -E "agg|index_scan|bn_|readout|head|gemm" gpurun_out/prof_r3h_b1024/summary.txt | cut -c1-60,112-190 | head -30
