mkdir -p gpurun_out; python -m pytest tests/test_gpu_fused_step.py -q -x 2>&1 | tail -15
EAGCN_BENCH_OVERLAP=0 bash tools/run_prof.sh r2_noovl --steps 20 --warmup 5 > gpurun_out/noovl_prof.log 2>&1
grep -v "^#" gpurun_out/prof_r2_noovl/timeline.txt | head -60
