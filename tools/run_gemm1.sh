mkdir -p gpurun_out/r2b
for T in 4809 19200 76800; do for cfg in auto 3 1; do
  if [ $cfg = auto ]; then T=$T python tools/gemm_bench.py; else EAGCN_GEMM_CFG=$cfg T=$T python tools/gemm_bench.py; fi
done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2b/gemm_bench.txt
