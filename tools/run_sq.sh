# usage: bash tools/run_sq.sh <outdir under gpurun_out> <command...>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
timeout 180 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $out/a -o a --output-format csv -- "$@" > $out/a.log 2>&1
timeout 180 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS -d $out/b -o b --output-format csv -- "$@" > $out/b.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/sq_counters.py $out/a gemm > $out/sq_a.txt 2>&1; python tools/sq_counters.py $out/b gemm > $out/sq_b.txt 2>&1
cat $out/sq_a.txt $out/sq_b.txt; tail -3 $out/a.log
rm -rf $out/a/*.db $out/b/*.db
