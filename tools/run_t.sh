mkdir -p gpurun_out/tests
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=300 -k "gcn" > gpurun_out/tests/pytest_gcn.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed|Error" gpurun_out/tests/pytest_gcn.log | tail -8
grep -E "gcn" gpurun_out/tests/pytest_gcn.log | tail -8 | cut -c1-260
