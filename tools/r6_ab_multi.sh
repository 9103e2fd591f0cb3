#!/bin/bash
# Round 6: same-box A/B of several BUILDS of the library: bash tools/r6_ab_multi.sh "<tag> <tag> ..." "<bench args>" ["<bench args>" ...]
# (eagcn_amd/lib/libeagcn_hip_<tag>.so built beforehand from source variants; `new` = the tree's own build)
R=$GRAFT_REPO_ROOT; L=$R/eagcn_amd/lib; out=$R/gpurun_out/r6q; mkdir -p $out; cd $R
tags=$1; shift
cp $L/libeagcn_hip.so $L/libeagcn_hip_new.so
one() { python bench.py --no-extras --no-cpu-baseline $1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_step']
        print('%-10s %-46s %.4f ms  ' % ('$TAG', '$1', d['ms_per_step']) + ' '.join('%s %.4f' % kv for kv in k.items() if kv[0] in ('agg', 'bn', 'head', 'gemm_pair', 'index')))
" | tee -a $out/ab_multi.txt; }
for rep in 1 2 3; do
  for TAG in $tags; do
    cp $L/libeagcn_hip_$TAG.so $L/libeagcn_hip.so
    for a in "$@"; do one "$a"; done
  done
done
cp $L/libeagcn_hip_new.so $L/libeagcn_hip.so
