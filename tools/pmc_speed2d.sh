#!/bin/bash
# PMC passes over the whole speed2d forward (eager launches): where do the small kernels spend their cycles?
#   gpurun -- 'bash tools/pmc_speed2d.sh'   -> gpurun_out/r05_pmc_speed2d_<group>.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rm -rf /tmp/pmc_s2d_$i
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_s2d_$i -o out -- python $R/bench.py --workload speed2d --steps 3 --warmup 1 --no-graph --no-predict --no-cpu-baseline > $R/gpurun_out/r05_pmc_speed2d_$i.log 2>&1
  DB=$(find /tmp/pmc_s2d_$i -name "*results.db" | head -1)
  python $R/tools/rocpd_stats.py pmc $DB > $R/gpurun_out/r05_pmc_speed2d_$i.txt 2>&1
done
