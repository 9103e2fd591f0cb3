#!/bin/bash
# Round-6 final evidence set (GPU box): full GPU test suite (parity report), the driver's bench command, the profile set of tools/run_r6_profiles.sh
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r6final; mkdir -p $out; cd $R
timeout 900 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; tail -1 $out/pytest.log | cut -c1-200; cp gpurun_out/parity_report.txt $out/parity_report.txt
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench.err; tail -c 600 $out/bench_default.json
timeout 300 python bench.py --optimizer flat --no-extras --no-cpu-baseline > $out/bench_train_step.json 2>> $out/bench.err
bash tools/run_r6_profiles.sh > $out/profiles.log 2>&1
ls gpurun_out/r6prof
