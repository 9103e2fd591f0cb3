#!/bin/bash
# rocprofv3 kernel trace of bench.py for one workload (run on the GPU box): per-kernel summary -> gpurun_out/<tag>_kernel_trace.txt
#   tools/trace_step.sh <workload> <batch> <tag> [extra bench args]
set -u
W=${1:-tox21_c2}; B=${2:-256}; TAG=${3:-trace}; shift 3 || true
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/$TAG
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/$TAG -o t -- python $ROOT/bench.py --workload $W --batch $B --steps 20 --warmup 5 --repeats 3 --no-extras --no-cpu-baseline "$@" > $ROOT/gpurun_out/$TAG/bench.log 2>&1
cd $ROOT
DB=$(find gpurun_out/$TAG -name "*results.db" | head -1)
python3 tools/rocprof_summary.py $DB > gpurun_out/${TAG}_kernel_trace.txt 2>&1
head -40 gpurun_out/${TAG}_kernel_trace.txt | cut -c1-60,105-190
find gpurun_out/$TAG -name "*.db" -size +20M -delete
