#!/bin/bash
# rocprofv3 counter passes over eager steps of one bench.py workload (run on the GPU box), every pass under its own `timeout`:
#   tools/prof_passes.sh <outdir under gpurun_out> "<passes: sq1 sq2 tcc fetch write>" <bench.py args ...>
# -> <outdir>/<pass>/t_counter_collection.csv + t_kernel_trace.csv, summary tables <outdir>/sq.txt, traffic.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; PASSES=$2; shift 2
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --repeats 1 --no-cpu-baseline --no-extras --eager $*"
run() { name=$1; shift; timeout ${PASS_TIMEOUT:-200} rocprofv3 --kernel-trace "$@" -d $OUT/$name -o t --output-format csv -- $CMD > $OUT/$name.log 2>&1 || echo "pass $name: rc $?"; }
for p in $PASSES; do
  case $p in
    sq1) run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS ;;
    sq2) run sq2 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SMEM ;;
    tcc) run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum ;;
    tcp) run tcp --pmc TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES TA_BUSY ;;
    fetch) run fetch --pmc FETCH_SIZE ;;
    write) run write --pmc WRITE_SIZE ;;
  esac
done
cd $R
for p in sq1 sq2; do [ -d $OUT/$p ] && python tools/sq_counters.py $OUT/$p > $OUT/$p.txt 2>&1; done
[ -d $OUT/fetch ] && [ -d $OUT/write ] && python tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/traffic > $OUT/traffic.log 2>&1
[ -d $OUT/tcc ] && python tools/pmc_generic.py $OUT/tcc > $OUT/tcc.txt 2>&1
[ -d $OUT/tcp ] && python tools/pmc_generic.py $OUT/tcp > $OUT/tcp.txt 2>&1
[ -d $OUT/fetch ] && python tools/pmc_generic.py $OUT/fetch > $OUT/fetch.txt 2>&1
[ -d $OUT/write ] && python tools/pmc_generic.py $OUT/write > $OUT/write.txt 2>&1
find $OUT -name '*.db' -delete; find $OUT -name '*_agent_info.csv' -delete
find $OUT -name '*.csv' -size +6M -delete
