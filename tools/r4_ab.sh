#!/bin/bash
# A/B of one environment switch over single-workload bench runs (GPU box): tools/r4_ab.sh VAR "v1 v2" "workload batch steps" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O; cd $R
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do
for w in "$@"; do
  read -r wl batch steps <<< "$w"
  for v in $VALS; do
    env $VAR=$v timeout 300 python bench.py --workload $wl --batch $batch --steps $steps --warmup 3 --no-extras --no-cpu-baseline > $O/${wl}_$v.json 2> $O/${wl}_$v.err
    python - $O/${wl}_$v.json $VAR=$v $wl <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('%-28s %-10s %.4f ms/step  ' % (sys.argv[2], sys.argv[3], d['ms_per_step']) + ' '.join('%s %.3f' % (n, v) for n, v in sorted(k.items(), key=lambda kv: -kv[1])[:8]))
PY
  done
done
done
