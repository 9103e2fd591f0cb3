#!/bin/bash
# Round 6: several 32-column chunks per lagg workgroup (EAGCN_LAGG_CPW / EAGCN_LAGG_CPW_BWD), whole step + kernel classes per workload
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r6cpw; mkdir -p $out; cd $R
run() { # tag env... -- bench args
  tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_ms_per_step']
        print('%-28s %-44s %.4f ms  agg %.4f bn %.4f gemm_pair %.4f' % ('$tag', '$*', d['ms_per_step'], k['agg'], k['bn'], k['gemm_pair']))
" | tee -a $out/ab.txt
}
for rep in 1 2; do
run base -- --batch 1024 --steps 30 --warmup 10 --repeats 5
run bwd5 EAGCN_LAGG_CPW=5 EAGCN_LAGG_CPW_BWD=1 -- --batch 1024 --steps 30 --warmup 10 --repeats 5
run bwd3 EAGCN_LAGG_CPW=3 EAGCN_LAGG_CPW_BWD=1 -- --batch 1024 --steps 30 --warmup 10 --repeats 5
run base -- --workload lipo_c4 --steps 20 --warmup 5 --repeats 5
run bwd4 EAGCN_LAGG_CPW=4 EAGCN_LAGG_CPW_BWD=1 -- --workload lipo_c4 --steps 20 --warmup 5 --repeats 5
run bwd8 EAGCN_LAGG_CPW=8 EAGCN_LAGG_CPW_BWD=1 -- --workload lipo_c4 --steps 20 --warmup 5 --repeats 5
run base -- --workload c5_synth --steps 5 --warmup 2 --repeats 3
run bwd4 EAGCN_LAGG_CPW=4 EAGCN_LAGG_CPW_BWD=1 -- --workload c5_synth --steps 5 --warmup 2 --repeats 3
run bwd8 EAGCN_LAGG_CPW=8 EAGCN_LAGG_CPW_BWD=1 -- --workload c5_synth --steps 5 --warmup 2 --repeats 3
done
for rep in 1 2; do
run hiv_nowfuse EAGCN_LAGG_WFUSE=0 -- --workload hiv_c3 --steps 6 --warmup 2 --repeats 3
run hiv_wfuse -- --workload hiv_c3 --steps 6 --warmup 2 --repeats 3
run hiv_wfuse_cpw4 EAGCN_LAGG_CPW=4 EAGCN_LAGG_CPW_BWD=1 -- --workload hiv_c3 --steps 6 --warmup 2 --repeats 3
run hiv_wfuse_cpw8 EAGCN_LAGG_CPW=8 EAGCN_LAGG_CPW_BWD=1 -- --workload hiv_c3 --steps 6 --warmup 2 --repeats 3
run hiv_wfuse_cpw8_fwd EAGCN_LAGG_CPW=8 EAGCN_LAGG_CPW_BWD=1 EAGCN_LAGG_FWD_MAXB=4096 -- --workload hiv_c3 --steps 6 --warmup 2 --repeats 3
done
