# Round-end verification on the GPU box: the whole -m gpu suite, smoke(), the default bench line (with both CPU baselines).
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/final/pytest.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/final/smoke.log
timeout 900 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/final/bench.json
