# Round-3 profile set (GPU box): bash tools/run_r3_profiles.sh  -> gpurun_out/r3prof/* (copied into profiles/ afterwards)
#  kernel trace + step timeline (graph replay) of the default command's workload and of batch 1024 / HIV; FETCH_SIZE / WRITE_SIZE
#  passes (eager launches, separate runs) with the tile-contiguous and with the XCD-local GEMM schedule; SQ counters of every kernel
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r3prof; mkdir -p $out
cd $R
bash tools/run_prof.sh r3_b256 --steps 20 --warmup 5 > $out/b256.log 2>&1
bash tools/run_prof.sh r3_b1024 --steps 20 --warmup 5 --batch 1024 > $out/b1024.log 2>&1
bash tools/run_prof.sh r3_hiv --steps 6 --warmup 2 --workload hiv_c3 > $out/hiv.log 2>&1
cd /tmp; export TMPDIR=/tmp
for xk in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    EAGCN_GEMM3_XK=$xk timeout 180 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_xk${xk}_$c -o t --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-extras --eager > $out/pmc_xk${xk}_$c.log 2>&1
  done
done
cd $R
python tools/pmc_traffic.py $out/pmc_xk0_FETCH_SIZE $out/pmc_xk0_WRITE_SIZE $out/pmc_traffic > $out/pmc.log 2>&1
python tools/pmc_traffic.py $out/pmc_xk1_FETCH_SIZE $out/pmc_xk1_WRITE_SIZE $out/pmc_traffic_xcd_local > $out/pmc_xk1.log 2>&1
o=$out/sq; mkdir -p $o; cd /tmp
timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $o/a -o a --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-extras --eager > $o/a.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $o/b -o b --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-extras --eager --batch 1024 > $o/b.log 2>&1
cd $R
python tools/sq_counters.py $o/a > $out/sq_b256.txt 2>&1; python tools/sq_counters.py $o/b > $out/sq_b1024.txt 2>&1
rm -rf $out/pmc_*/*.db $out/pmc_*/*/*.db $o/a/*.db $o/b/*.db $o/a/*/*.db $o/b/*/*.db
find $out -name '*_agent_info.csv' -delete; find $out -name '*counter_collection.csv' -size +8M -delete; find $out -name '*kernel_trace.csv' -size +8M -delete
du -sh $out; head -12 $out/pmc_traffic.txt; head -4 $out/pmc_traffic_xcd_local.txt; head -14 $out/sq_b256.txt; head -14 $out/sq_b1024.txt
