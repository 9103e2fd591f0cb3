#!/bin/bash
# kernel traces of three workloads with the LDS-staged aggregation (EAGCN_AGG=lds) and the dense one
for cfg in ${CFGS:-"hiv_c3 1024" "c5_synth 1024"}; do
  set -- $cfg
  for agg in ${AGGS:-lds dense}; do
    EAGCN_AGG=$agg bash tools/trace_step.sh $1 $2 r5f_${1}_${2}_$agg > /dev/null 2>&1
    echo "==== $1 B=$2 EAGCN_AGG=$agg"; grep -E "ms_per_step|\"value\"" gpurun_out/r5f_${1}_${2}_$agg/bench.log | tail -1 | cut -c1-200
    head -12 gpurun_out/r5f_${1}_${2}_${agg}_kernel_trace.txt | cut -c1-70,105-175
    rm -rf gpurun_out/r5f_${1}_${2}_$agg
  done
done
