#!/bin/bash
# ablation probes of the 256 x 128 kernel (EAGCN_BX3W_DBG: 1 no LDS-DMA | 2 no MFMA | 3 no fragment reads) at two large shapes
OUT=gpurun_out/r5b; mkdir -p $OUT
for d in 0 1 2 3; do
  echo "==== EAGCN_BX3W_DBG=$d"
  EAGCN_BX3W_DBG=$d timeout 200 tools/bx3_bench one 19200 400 720 10 2>&1 | grep -E "forward|dX|dW"
  EAGCN_BX3W_DBG=$d timeout 200 tools/bx3_bench one 100000 512 1024 6 2>&1 | grep -E "forward|dX|dW"
done 2>&1 | tee $OUT/ablation.txt
