mkdir -p gpurun_out/x3
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_dist.py -m gpu -q -x --timeout=300 > gpurun_out/x3/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/x3/pytest.log | tail -5
bash tools/run_prof.sh x3_b1024 --batch 1024 --steps 20 > gpurun_out/x3/prof_b1024.txt 2>&1
grep -h "head_\|readout\|pack_params" gpurun_out/prof_x3_b1024/summary.txt | cut -c1-180
timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 1024 --steps 30 > gpurun_out/x3/b1024.json 2>/dev/null; cut -c1-260 gpurun_out/x3/b1024.json
timeout 200 python bench.py --workload hiv_c3 --no-cpu-baseline --no-extras --repeats 3 --steps 10 > gpurun_out/x3/hiv.json 2>gpurun_out/x3/hiv.err; cut -c1-200 gpurun_out/x3/hiv.json
timeout 200 python bench.py --workload lipo_c4 --no-cpu-baseline --no-extras --repeats 3 --steps 20 > gpurun_out/x3/lipo.json 2>gpurun_out/x3/lipo.err; cut -c1-200 gpurun_out/x3/lipo.json
