mkdir -p gpurun_out/x5
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_fused_step.py -m gpu -q -x --timeout=300 > gpurun_out/x5/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/x5/pytest.log | tail -5
bash tools/run_prof.sh x5_b1024 --batch 1024 --steps 20 > gpurun_out/x5/prof_b1024.txt 2>&1
grep -h "head_\|readout\|pack_params" gpurun_out/prof_x5_b1024/summary.txt | cut -c1-180
bash tools/run_prof.sh x5_b256 --steps 30 > gpurun_out/x5/prof_b256.txt 2>&1
grep -h "head_\|readout\|pack_params" gpurun_out/prof_x5_b256/summary.txt | cut -c1-180
timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 1024 --steps 30 > gpurun_out/x5/b1024.json 2>/dev/null; cut -c1-260 gpurun_out/x5/b1024.json
timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/x5/b256.json 2>/dev/null; cut -c1-260 gpurun_out/x5/b256.json
