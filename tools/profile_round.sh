#!/bin/bash
# Round profile on the GPU box (gpurun): rocprofv3 kernel stats of the contract bench (1 and 2 graph branches) and PMC
# passes (separate runs, one counter group each) of the dominant GEMM instantiations on the dominant shape.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/profile_round.sh r03'      (SKIP_PMC=1: kernel stats only)
# Summaries land in gpurun_out/<tag>_*; copy the ones to keep into profiles/.
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
[ -z "$SKIP_PMC" ] && rm -f $OUT/${TAG}_pmc_summary.txt $OUT/${TAG}_pmc_summary_bf16x3.txt
rm -f $OUT/${TAG}_tune*.json
B="--no-cpu-baseline --no-predict --no-clip-leg --no-bf16x3 --tune-cache $OUT/${TAG}_tune.json"
python bench.py $B --steps 10 > $OUT/${TAG}_b0.log 2>&1
for S in 1 2; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_s$S -o out -- python $GRAFT_REPO_ROOT/bench.py $B --steps 20 --warmup 3 --streams $S > $OUT/${TAG}_prof_s$S.log 2>&1)
  DB=$(ls $OUT/${TAG}_prof_s$S/*/*results.db $OUT/${TAG}_prof_s$S/*results.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py kernels $DB $OUT/${TAG}_bench_kernel_stats_streams$S.csv
done
if [ -z "$SKIP_PMC" ]; then
# PMC: gemm1x1 <2,2,2,3> (cfg 9), <4,1,1,3> (11), <4,1,1,2> (12), <4,1,1,1> (13), pre_relu = 0 -> the <.., false, false, false> instantiations
for CFG in 9 11 12 13; do
 for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace -d $OUT/${TAG}_pmc_${CFG}_$C -o out -- python $GRAFT_REPO_ROOT/tools/bench_one.py 32 576 576 1 1 $CFG 4 0 > /dev/null 2>&1)
  DB=$(ls $OUT/${TAG}_pmc_${CFG}_$C/*/*results.db $OUT/${TAG}_pmc_${CFG}_$C/*results.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py pmc $DB gemm1x1 >> $OUT/${TAG}_pmc_summary.txt
 done
done
fi
# split-bf16 mode: kernel stats of the same bench in bf16x3 mode (one graph branch) and the PMC passes of the wide tiling
B3="--gemm bf16x3 --no-cpu-baseline --no-predict --no-clip-leg --no-bf16x3 --tune-cache $OUT/${TAG}_tune_bf16x3.json"
python bench.py $B3 --steps 10 > $OUT/${TAG}_b3.log 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof_b3 -o out -- python $GRAFT_REPO_ROOT/bench.py $B3 --steps 20 --warmup 3 --streams 1 > $OUT/${TAG}_prof_b3.log 2>&1)
DB=$(ls $OUT/${TAG}_prof_b3/*/*results.db $OUT/${TAG}_prof_b3/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py kernels $DB $OUT/${TAG}_bench_kernel_stats_bf16x3_streams1.csv
[ -z "$SKIP_PMC" ] && for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace -d $OUT/${TAG}_pmc_w_$C -o out -- python $GRAFT_REPO_ROOT/tools/bench_one.py 32 576 576 1 1 14 4 0 1 > /dev/null 2>&1)
  DB=$(ls $OUT/${TAG}_pmc_w_$C/*/*results.db $OUT/${TAG}_pmc_w_$C/*results.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py pmc $DB gemm1x1s_wide >> $OUT/${TAG}_pmc_summary_bf16x3.txt
done
[ -z "$SKIP_PMC" ] && cat $OUT/${TAG}_pmc_summary.txt $OUT/${TAG}_pmc_summary_bf16x3.txt
tail -1 $OUT/${TAG}_prof_b3.log | cut -c1-300
tail -1 $OUT/${TAG}_prof_s1.log | cut -c1-400
head -12 $OUT/${TAG}_bench_kernel_stats_streams1.csv
[ -z "$SKIP_PMC" ] && python tools/make_pmc_json.py $OUT/${TAG}_pmc_summary.txt $OUT/${TAG}_pmc_summary_bf16x3.txt > $OUT/${TAG}_pmc_dominant_kernel.json
# keep the merged-back payload small
rm -rf $OUT/${TAG}_prof_s1 $OUT/${TAG}_prof_s2 $OUT/${TAG}_prof_b3 $OUT/${TAG}_pmc_[0-9]* $OUT/${TAG}_pmc_w_*
