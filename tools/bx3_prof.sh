#!/bin/bash
# rocprofv3 passes over one layer shape of tools/bx3_bench (run on the GPU box): kernel trace, SQ counters, cache counters.
#   tools/bx3_prof.sh <T> <FIN> <FP> <outdir> [passes: trace sq1 sq2 tcp tcc fetch]
# every pass runs under its own `timeout` (a pass with an unknown counter name hung for 10 minutes once)
set -u
T=${1:-19200}; FIN=${2:-400}; FP=${3:-720}; OUT=${4:-gpurun_out/bx3_prof}; PASSES=${5:-"trace sq1 sq2 tcc fetch"}
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp
CMD="$ROOT/tools/bx3_bench one $T $FIN $FP 3"
run() { name=$1; shift; timeout 120 rocprofv3 --kernel-trace "$@" -d $ROOT/$OUT/$name -o t --output-format csv -- $CMD > $ROOT/$OUT/$name.log 2>&1 || echo "pass $name: rc $?"; }
for p in $PASSES; do
  case $p in
    trace) run trace --stats ;;
    sq1) run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES ;;
    sq2) run sq2 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL ;;
    tcp) run tcp --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_READ_TAGCONFLICT_STALL_CYCLES TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_READ_LDS_WAVEFRONTS ;;
    tcc) run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum ;;
    fetch) run fetch --pmc FETCH_SIZE ;;
  esac
done
cd $ROOT
python3 tools/bx3_prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
