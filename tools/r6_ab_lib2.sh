#!/bin/bash
# like r6_ab_lib.sh, all five workloads, two repetitions
R=$GRAFT_REPO_ROOT; L=$R/eagcn_amd/lib; out=$R/gpurun_out/r6q; mkdir -p $out; cd $R
cp $L/libeagcn_hip.so $L/libeagcn_hip_new.so
one() { python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_step']
        print('%-5s %-46s %.4f ms  ' % ('$TAG', '$*', d['ms_per_step']) + ' '.join('%s %.4f' % kv for kv in k.items() if kv[0] in ('agg', 'bn', 'head', 'gemm_pair', 'index')))
" | tee -a $out/ab_lib.txt; }
for rep in 1 2; do
  for TAG in new base; do
    cp $L/libeagcn_hip_$TAG.so $L/libeagcn_hip.so
    one --steps 50 --warmup 10 --repeats 9
    one --batch 1024 --steps 30 --warmup 10 --repeats 5
    one --workload lipo_c4 --steps 20 --warmup 5 --repeats 5
    one --workload hiv_c3 --steps 6 --warmup 2 --repeats 3
    one --workload c5_synth --steps 5 --warmup 2 --repeats 3
  done
done
cp $L/libeagcn_hip_new.so $L/libeagcn_hip.so
