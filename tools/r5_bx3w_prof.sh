#!/bin/bash
# counters of the forward product at one large shape, both kernels: where do the operand bytes come from?
set -u
OUT=gpurun_out/r5c; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
T=${1:-100000}; FIN=${2:-512}; FP=${3:-1024}
cd /tmp
for w in 0 1; do
  CMD="$ROOT/tools/bx3_bench fwd $T $FIN $FP 3 $w"
  run() { name=$1; shift; timeout 150 rocprofv3 --kernel-trace "$@" -d $ROOT/$OUT/w$w/$name -o t --output-format csv -- $CMD > $ROOT/$OUT/w$w.$name.log 2>&1 || echo "pass $name: rc $?"; }
  mkdir -p $ROOT/$OUT/w$w
  run trace --stats
  run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
  run sq2 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
  run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
  run fetch --pmc FETCH_SIZE
  run tcp --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_READ_LDS_WAVEFRONTS
  run grbm --pmc GRBM_GUI_ACTIVE
  cd $ROOT; python3 tools/bx3_prof_summary.py $OUT/w$w > $OUT/summary_w$w.txt 2>&1; cd /tmp
done
cd $ROOT
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/summary_w0.txt $OUT/summary_w1.txt
