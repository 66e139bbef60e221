#!/bin/bash
# scripts/pmc_issue.sh <bench args...> -- instruction mix and issue-unit activity per kernel symbol, one bench frame (inside gpurun):
# vector / scalar / branch / LDS instruction counts and the cycles the scalar unit and the vector ALUs were active
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES"; do
  rm -rf gpurun_out/pmc_issue
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_issue -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timers "$@" > /dev/null 2>&1
  python3 scripts/pmc_sum.py gpurun_out/pmc_issue | grep "k_trace\|k_shade"
done | tee gpurun_out/r03_issue_counters.txt
