# usage: bash tools/run_prof.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>/{summary,timeline}.txt
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --memory-copy-trace -d $out -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --repeats 2 "$@" > $out/bench.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(ls $out/*.db $out/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $db > $out/summary.txt 2>&1
python tools/rocprof_timeline.py $db > $out/timeline.txt 2>&1
rm -f $db
head -45 $out/summary.txt; cat $out/timeline.txt; tail -2 $out/bench.log | cut -c1-300
