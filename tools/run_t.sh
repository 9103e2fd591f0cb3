mkdir -p gpurun_out/tests
timeout 300 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout=200 > gpurun_out/tests/pytest_bf16.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed|^E  |gemm mode" gpurun_out/tests/pytest_bf16.log | tail -8 | cut -c1-300
