#!/bin/bash
# Round-6 final evidence for the LAST build (the GPU suite ran in the call before): the driver's bench command, the training iteration, the profile set
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r6final; mkdir -p $out; cd $R
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench.err; tail -c 400 $out/bench_default.json
timeout 300 python bench.py --optimizer flat --no-extras --no-cpu-baseline > $out/bench_train_step.json 2>> $out/bench.err
bash tools/run_r6_profiles.sh > $out/profiles.log 2>&1
ls gpurun_out/r6prof
