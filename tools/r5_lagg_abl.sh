#!/bin/bash
# ablation of lagg.hip (EAGCN_LAGG_DBG bits: 1 no S phase, 2 no gathers, 4 no stores): tools/r5_lagg_abl.sh [workload batch]
W=${1:-c5_synth}; B=${2:-1024}
for d in 0 1 2 4 7; do
  EAGCN_LAGG_DBG=$d EAGCN_AGG=lds bash tools/trace_step.sh $W $B r5g_$d > /dev/null 2>&1
  echo "==== EAGCN_LAGG_DBG=$d"; grep "lagg_kernel" gpurun_out/r5g_${d}_kernel_trace.txt | cut -c1-40,105-175
  rm -rf gpurun_out/r5g_$d
done
