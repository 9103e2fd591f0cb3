#!/bin/bash
# Round-6 profile set (GPU box): bash tools/run_r6_profiles.sh -> gpurun_out/r6prof/* (the summaries are copied into profiles/r06_* by hand)
#  kernel traces + one-step timelines (graph replay) of the default command and of B = 1024 / HIV / Lipo / C5; SQ counters and
#  FETCH_SIZE / WRITE_SIZE (separate passes, eager launches) of the default workload; SQ + FETCH / WRITE of C5 (bx3w, lagg) and HIV
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r6prof; mkdir -p $out
cd $R
bash tools/run_prof.sh r6_b256 --steps 20 --warmup 5 > $out/b256.log 2>&1
bash tools/run_prof.sh r6_b1024 --steps 20 --warmup 5 --batch 1024 > $out/b1024.log 2>&1
bash tools/run_prof.sh r6_hiv --steps 6 --warmup 2 --workload hiv_c3 > $out/hiv.log 2>&1
bash tools/run_prof.sh r6_lipo --steps 10 --warmup 3 --workload lipo_c4 > $out/lipo.log 2>&1
bash tools/run_prof.sh r6_c5 --steps 3 --warmup 2 --workload c5_synth > $out/c5.log 2>&1
PASS_TIMEOUT=200 bash tools/prof_passes.sh r6prof/b256 "sq1 sq2 fetch write" --steps 10 --warmup 3
PASS_TIMEOUT=240 bash tools/prof_passes.sh r6prof/c5 "sq1 fetch write" --workload c5_synth --batch 1024 --steps 2 --warmup 1
PASS_TIMEOUT=240 bash tools/prof_passes.sh r6prof/hiv "sq1 fetch write" --workload hiv_c3 --batch 1024 --steps 3 --warmup 1
for t in b256 b1024 hiv lipo c5; do cp $R/gpurun_out/prof_r6_$t/summary.txt $out/${t}_kernel_trace.txt 2>/dev/null; cp $R/gpurun_out/prof_r6_$t/timeline.txt $out/${t}_timeline.txt 2>/dev/null; done
rm -rf $R/gpurun_out/prof_r6_*
find $out -name '*.csv' -delete
du -sh $out; head -30 $out/b256_kernel_trace.txt; cat $out/b256/traffic.txt | head -14; head -8 $out/b256/sq1.txt
# round 6: the two queues of three consecutive steps (main stream / batch-preparation stream), configs[1] and B = 1024
bash tools/r6_streams.sh > /dev/null 2>&1; cp $R/gpurun_out/prof_r6s/streams.txt $out/b256_streams.txt 2>/dev/null
