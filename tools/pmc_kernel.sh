#!/bin/bash
# SQ issue/stall counters of one GEMM instantiation on one shape (gpurun):
#   bash tools/pmc_kernel.sh <tag> <kernel substring> -- <bench_one.py arguments>
# One rocprofv3 pass per counter group (PMC passes carry --kernel-trace only); falls back to one pass per counter
# when a group does not fit the hardware.
TAG=$1; SUB=$2; shift 3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
SUM=$OUT/${TAG}_pmc.txt
: > $SUM
run() {   # counters...
  local D=$OUT/${TAG}_pmc_tmp
  rm -rf $D
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $D -o out -- python $GRAFT_REPO_ROOT/tools/bench_one.py $ARGS > $D.log 2>&1)
  local DB=$(ls $D/*/*results.db $D/*results.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py pmc $DB $SUB >> $SUM && rm -rf $D $D.log && return 0
  rm -rf $D; return 1
}
ARGS="$@"
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  run $G || for C in $G; do run $C || echo "$C unavailable" >> $SUM; done
done
cat $SUM
