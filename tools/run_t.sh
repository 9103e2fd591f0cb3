timeout 200 bash tools/run_prof.sh r2_sparse --steps 20 --warmup 5 > gpurun_out/sparse_prof.log 2>&1
grep -v "^#" gpurun_out/prof_r2_sparse/timeline.txt | grep -E "sagg|csr|scan|bn_|gemm"
tail -1 gpurun_out/prof_r2_sparse/timeline.txt
