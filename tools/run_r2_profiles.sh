# Round-2 profile set: kernel trace + step timeline (graph replay) for the headline workload and B=1024,
# FETCH_SIZE / WRITE_SIZE passes (eager launches, separate runs), SQ counters of the GEMM kernels.
# usage (GPU box): bash tools/run_r2_profiles.sh   -> gpurun_out/r2prof/*
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r2prof; mkdir -p $out
cd $R
bash tools/run_prof.sh r2_b256 --steps 20 --warmup 5 > $out/b256.log 2>&1
bash tools/run_prof.sh r2_b1024 --steps 20 --warmup 5 --batch 1024 > $out/b1024.log 2>&1
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 180 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_$c -o t --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-extras --eager > $out/pmc_$c.log 2>&1
done
cd $R
python tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_traffic > $out/pmc.log 2>&1
bash tools/run_sq.sh r2prof/sq python $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-extras --eager > $out/sq.log 2>&1
rm -rf $out/pmc_*/*.db $out/pmc_*/*/*.db
find $out -name '*_agent_info.csv' -delete
du -sh $out; tail -5 $out/pmc.log; tail -12 $out/sq.log
bash tools/run_prof.sh r2_hiv --steps 6 --warmup 2 --workload hiv_c3 > $out/hiv.log 2>&1
