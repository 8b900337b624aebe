set -x
cd $GRAFT_REPO_ROOT
python bench.py --tune-cache gpurun_out/tune.json --no-cpu-baseline --steps 10 > gpurun_out/b0.log 2>&1
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_s2 -o out -- python $GRAFT_REPO_ROOT/bench.py --tune-cache $GRAFT_REPO_ROOT/gpurun_out/tune.json --no-cpu-baseline --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_s2.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_s1 -o out -- python $GRAFT_REPO_ROOT/bench.py --tune-cache $GRAFT_REPO_ROOT/gpurun_out/tune.json --no-cpu-baseline --steps 20 --warmup 3 --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/prof_s1.log 2>&1)
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
 (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmcf_$c -o out -- python $GRAFT_REPO_ROOT/tools/bench_one.py 32 576 576 1 1 11 4 > /dev/null 2>&1)
 (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmcg_$c -o out -- python $GRAFT_REPO_ROOT/tools/bench_one.py 32 576 576 1 1 12 4 > /dev/null 2>&1)
done
ls gpurun_out/prof_s2 gpurun_out/prof_s1 | head; tail -1 gpurun_out/prof_s2.log | cut -c1-300; tail -1 gpurun_out/prof_s1.log | cut -c1-300
