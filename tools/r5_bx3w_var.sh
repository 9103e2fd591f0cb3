#!/bin/bash
# where the LDS-DMA requests go inside the 256 x 128 kernel's k-tile (EAGCN_BX3W_VAR: bit 0 s_setprio, bits 1.. DMAPOS)
OUT=gpurun_out/r5d; mkdir -p $OUT
for v in ${VARS:-0 2 4 3 5}; do
  echo "==== EAGCN_BX3W_VAR=$v"
  EAGCN_BX3W_VAR=$v timeout 200 tools/bx3_bench one 19200 400 720 10 2>&1 | grep -E "forward|dX|dW"
  EAGCN_BX3W_VAR=$v timeout 200 tools/bx3_bench one 100000 512 1024 6 2>&1 | grep -E "forward|dX|dW"
  EAGCN_BX3W_VAR=$v timeout 200 tools/bx3_bench one 25000 512 6320 4 2>&1 | grep -E "forward|dX|dW"
done 2>&1 | tee $OUT/variants.txt
EAGCN_BX3W_VAR=4 timeout 300 tools/bx3_bench check > $OUT/check_v4.txt 2>&1; tail -1 $OUT/check_v4.txt
EAGCN_BX3W_VAR=2 timeout 300 tools/bx3_bench check > $OUT/check_v2.txt 2>&1; tail -1 $OUT/check_v2.txt
