#!/bin/bash
# parity subset + per-workload step times and kernel classes (GPU box); EAGCN_* switches can be given per run via RUNS
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/aggab; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
for w in "c5_synth 1024 6" "hiv_c3 1024 10" "tox21_c2 1024 30" "lipo_c4 512 30"; do
    set -- $w
    timeout 300 python bench.py --workload $1 --batch $2 --steps $3 --warmup 3 --no-extras --no-cpu-baseline > $O/${1}.json 2> $O/${1}.err
    python - $O/${1}.json $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('%-10s %.4f ms/step  ' % (sys.argv[2], d['ms_per_step']) + ' '.join('%s %.3f' % (n, v) for n, v in sorted(k.items(), key=lambda kv: -kv[1])[:9]))
PY
done
PASS_TIMEOUT=240 bash tools/prof_passes.sh aggab/c5 "sq1" --workload c5_synth --batch 1024 --steps 2 --warmup 1
grep -h "agg\|kernel \|index" $O/c5/sq1.txt | head -8
PASS_TIMEOUT=240 bash tools/prof_passes.sh aggab/hiv "sq1" --workload hiv_c3 --batch 1024 --steps 3 --warmup 1
grep -h "agg\|kernel \|index\|bx3\|bn_" $O/hiv/sq1.txt | head -16
