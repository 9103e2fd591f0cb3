mkdir -p gpurun_out/tests
timeout 300 python -m pytest tests/test_gpu_general_relations.py tests/test_gpu_collate.py -m gpu -q --timeout=200 > gpurun_out/tests/pytest_gen.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed|^E  " gpurun_out/tests/pytest_gen.log | tail -14 | cut -c1-300
