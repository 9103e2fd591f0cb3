mkdir -p gpurun_out; python -m pytest tests/test_gpu_training.py -q -x 2>&1 | tail -15
bash tools/run_r2_profiles.sh
