mkdir -p gpurun_out/tests
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=200 -k "gat" > gpurun_out/tests/pytest_gat.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed|^E  " gpurun_out/tests/pytest_gat.log | tail -12 | cut -c1-300
