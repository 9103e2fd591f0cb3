#!/bin/bash
# round 5: the 256 x 128 plane-GEMM kernel (csrc/gemm_bx3w.hip) against the 128 x 128 one -- accuracy, timings, priority A/B
OUT=gpurun_out/r5a; mkdir -p $OUT
timeout 300 tools/bx3_bench check > $OUT/check.txt 2>&1; echo "check rc=$?"; tail -3 $OUT/check.txt
timeout 300 tools/bx3_bench time 10 > $OUT/time_prio0.txt 2>&1; echo "time rc=$?"
EAGCN_BX3W_PRIO=1 timeout 300 tools/bx3_bench time 10 > $OUT/time_prio1.txt 2>&1; echo "time prio1 rc=$?"
cat $OUT/time_prio0.txt; echo ---- prio 1; grep -A5 "" $OUT/time_prio1.txt | grep -E "^[a-z]|forward|dX \+" 
timeout 900 python -m pytest tests/test_gpu_bx3.py -x -q --timeout=600 > $OUT/pytest_bx3.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_bx3.log
