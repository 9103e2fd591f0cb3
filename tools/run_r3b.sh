# round 3, call B: clamp-free interleaved main loop of gemm3 (micro + step), gemm + training tests
mkdir -p gpurun_out/r3b
for t in 4809 19200; do
  T=$t timeout 300 python tools/gemm_sk_bench.py > gpurun_out/r3b/gemm_T$t.txt 2>&1
done
T=25000 FIN=2512 FP=6320 CHECK=0 timeout 300 python tools/gemm_sk_bench.py > gpurun_out/r3b/gemm_hiv.txt 2>&1
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py tests/test_gpu_fused_step.py -m gpu -q --timeout=300 -x -k "gemm or fifty or fused or compact or tox21" -s > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed|trajectory parity" gpurun_out/r3b/pytest.log | tail -8
cat gpurun_out/r3b/gemm_*.txt | tr '|' '\n' | grep -E "pair|T=|fwd|dX|dW"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3b/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'avg_us', d['roofline']['avg_launch_us'], d['kernel_ms_per_step'], {k:(v.get('ms_per_step'),v.get('step_frac')) for k,v in d.get('extra',{}).items()})
PY
