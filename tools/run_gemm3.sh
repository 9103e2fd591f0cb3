mkdir -p gpurun_out/r2e
( for T in 4809 19200; do for W in 256; do
   EAGCN_GEMM3_WGS=$W T=$T timeout 120 python tools/gemm2_bench.py
 done; done
 EAGCN_GEMM3_WGS=256 T=76800 CHECK=0 timeout 120 python tools/gemm2_bench.py
 EAGCN_GEMM3_WGS=256 T=25000 FIN=512 FP=6320 CHECK=0 timeout 120 python tools/gemm2_bench.py
 EAGCN_GEMM3_WGS=256 T=262144 FIN=512 FP=1024 CHECK=0 timeout 120 python tools/gemm2_bench.py
 EAGCN_GEMM3_WGS=512 T=19200 timeout 120 python tools/gemm2_bench.py
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2e/gemm3_bench_b.txt
