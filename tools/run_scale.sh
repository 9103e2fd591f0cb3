#!/bin/bash
# The data-parallel scaling curve in one command, for the first lease on an 8-GPU MI355X node (the driver's
# SCALE_rNN.json covers bench.py's default workload only).
#   tools/run_scale.sh [out_dir]            (GPUS="1 2 4 8" WORKLOADS="tox21_c2:1024 lipo_c4:512 c5_synth:1024" to override)
# For every workload x rank count: bench.py under torch.distributed.run (one rank per GPU over RCCL), --require-in-graph-allreduce
# (the run FAILS instead of silently timing a host-issued collective), then one summary line per run:
#   workload  n_gpus  molecules/s  ms_per_step  per-rank ms  rccl_world  all-reduce form
# and the weak-scaling efficiency against the same workload's 1-GPU line.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/scale}
GPUS=${GPUS:-"1 2 4 8"}
WORKLOADS=${WORKLOADS:-"tox21_c2:1024 lipo_c4:512 c5_synth:1024"}
STEPS=${STEPS:-30}
WARMUP=${WARMUP:-8}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
port=29611
for wl in $WORKLOADS; do
    name=${wl%%:*}; batch=${wl##*:}
    for n in $GPUS; do
        log="$OUT/${name}_b${batch}_n${n}"
        port=$((port + 1))
        if [ "$n" = 1 ]; then
            timeout 900 python bench.py --gpus 1 --workload "$name" --batch "$batch" --steps "$STEPS" --warmup "$WARMUP" \
                --no-extras --no-cpu-baseline > "$log.json" 2> "$log.err"
        else
            timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" \
                bench.py --gpus "$n" --workload "$name" --batch "$batch" --steps "$STEPS" --warmup "$WARMUP" \
                --no-extras --no-cpu-baseline --require-in-graph-allreduce > "$log.json" 2> "$log.err"
        fi
        rc=$?
        [ $rc -ne 0 ] && echo "$name n=$n: bench.py exited $rc (see $log.err)" && tail -3 "$log.err"
    done
done
python - "$OUT" <<'PY'
import glob, json, os, sys
rows = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], '*.json'))):
    line = next((l for l in open(f) if l.startswith('{')), None)
    if line is None:
        continue
    d = json.loads(line)
    rows.setdefault(d['config']['workload'], {})[d['n_gpus']] = d
for wl, by_n in rows.items():
    base = by_n.get(1)
    for n in sorted(by_n):
        d = by_n[n]
        eff = '' if base is None else ' efficiency %.3f' % (d['value'] / (n * base['value']))
        print('%-34s n=%d  %12.1f %s  %8.4f ms/step  ranks %s  rccl_world %s  %s%s' % (
            wl, n, d['value'], d['unit'], d['ms_per_step'], d.get('ms_per_step_by_rank'), d.get('rccl_world', '-'),
            d.get('gradient_allreduce', 'no collective (1 rank)'), eff))
PY
