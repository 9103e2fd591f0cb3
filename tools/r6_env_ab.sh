#!/bin/bash
# Round 6: A/B of environment switches inside one call: bash tools/r6_env_ab.sh "<bench args>" "ENV1=a ENV2=b" "ENV3=c" ...  (first variant: none)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r6q; mkdir -p $out; cd $R
args=$1; shift
one() { env $1 python bench.py --no-extras --no-cpu-baseline $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_step']
        print('%-44s %-40s %.4f ms  ' % ('$1', '$args', d['ms_per_step']) + ' '.join('%s %.4f' % kv for kv in k.items() if kv[0] in ('agg', 'bn', 'head', 'gemm', 'gemm_pair', 'index')))
" | tee -a $out/env_ab.txt; }
for rep in 1 2; do
  one X=1
  for v in "$@"; do one "$v"; done
done
