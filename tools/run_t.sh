mkdir -p gpurun_out/tests
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=200 -k "golden and not compact" > gpurun_out/tests/pytest_h.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/tests/pytest_h.log | tail -4
timeout 200 python bench.py --no-cpu-baseline --no-extras --repeats 9 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
