#!/bin/bash
# Round 6: one line per workload (whole step + kernel classes), graph replay: bash tools/r6_quick.sh <tag> [env...]
R=$GRAFT_REPO_ROOT; tag=$1; shift; out=$R/gpurun_out/r6q; mkdir -p $out; cd $R
run() {
  env "${ENVS[@]}" python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_step']
        print('%-14s %-50s %.4f ms (%.4f-%.4f)  ' % ('$tag', '$*', d['ms_per_step'], d['n_gpus'] * (d.get('config', {}).get('global_batch', 0)) / d['value_max'] * 1e3 if d.get('value_max') else 0, d['n_gpus'] * (d.get('config', {}).get('global_batch', 0)) / d['value_min'] * 1e3 if d.get('value_min') else 0) + ' '.join('%s %.4f' % kv for kv in k.items()))
" | tee -a $out/quick.txt
}
ENVS=("$@"); [ ${#ENVS[@]} -eq 0 ] && ENVS=(X=1)
run --steps 50 --warmup 10 --repeats 9
run --batch 1024 --steps 30 --warmup 10 --repeats 5
run --workload hiv_c3 --steps 6 --warmup 2 --repeats 3
run --workload lipo_c4 --steps 20 --warmup 5 --repeats 5
run --workload c5_synth --steps 5 --warmup 2 --repeats 3
