mkdir -p gpurun_out/tests
python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/tests/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/tests/pytest.log | tail -15
