#!/bin/bash
# where does the LDS-staged bond-list aggregation (csrc/lagg.hip) pay when it only runs in the BACKWARD (its lists then have the whole
# forward to arrive from the side stream, and it absorbs bn_bwd_apply)?  EAGCN_AGG=lds forces it, EAGCN_LAGG_PARTS bit 0 forward / bit 1 backward
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O; cd $R
for rep in 1 2; do
for w in "$@"; do
  read -r wl batch steps <<< "$w"
  for v in dense lds:2 lds:3; do
    agg=${v%%:*}; parts=${v##*:}; [ "$agg" = dense ] && parts=3
    EAGCN_AGG=$agg EAGCN_LAGG_PARTS=$parts timeout 300 python bench.py --workload $wl --batch $batch --steps $steps --warmup 3 --no-extras --no-cpu-baseline > $O/${wl}_$v.json 2> $O/${wl}_$v.err
    python - $O/${wl}_$v.json $v $wl $batch <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('%-8s %-10s B=%-5s %.4f ms/step  ' % (sys.argv[2], sys.argv[3], sys.argv[4], d['ms_per_step']) + ' '.join('%s %.3f' % (n, v) for n, v in sorted(k.items(), key=lambda kv: -kv[1])[:8]))
PY
  done
done
done
