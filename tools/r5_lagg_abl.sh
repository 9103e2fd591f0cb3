#!/bin/bash
# ablation of lagg.hip at the C5 shape (EAGCN_LAGG_DBG bits: 1 no S phase, 2 no gathers, 4 no stores, 8 dead rows loop)
for d in 0 1 2 4 7; do
  EAGCN_LAGG_DBG=$d EAGCN_AGG=lds bash tools/trace_step.sh c5_synth 1024 r5g_$d > /dev/null 2>&1
  echo "==== EAGCN_LAGG_DBG=$d"; grep "lagg_kernel" gpurun_out/r5g_${d}_kernel_trace.txt | cut -c1-40,80-160
  rm -rf gpurun_out/r5g_$d
done
