mkdir -p gpurun_out/tests
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_training.py -m gpu -q --timeout=300 > gpurun_out/tests/pytest_parity.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/tests/pytest_parity.log | tail -12
grep -E "AssertionError: \(" gpurun_out/tests/pytest_parity.log | head -20
