mkdir -p gpurun_out/tests
timeout 300 python -m pytest tests/test_gpu_nbucket.py -m gpu -q --timeout=200 > gpurun_out/tests/pytest_nb.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed|^E  " gpurun_out/tests/pytest_nb.log | tail -14 | cut -c1-300
