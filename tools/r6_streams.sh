out=$GRAFT_REPO_ROOT/gpurun_out/prof_r6s; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace -d $out -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --repeats 2 > $out/bench.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(ls $out/*.db $out/*/*.db 2>/dev/null | head -1)
python tools/rocprof_streams.py $db 3 > $out/streams.txt 2>&1
sqlite3 $db "pragma table_info(kernels)" > $out/cols.txt 2>&1
rm -f $db
