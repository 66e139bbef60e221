#!/bin/bash
# scripts/pmc_cache.sh <bench args...> -- one TA / TCP / TCC counter pass over one bench frame (inside gpurun): gpurun_out/r03_cache_counters.txt
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_cache
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_cache -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timers "$@" > /dev/null 2>&1
python3 scripts/pmc_sum.py gpurun_out/pmc_cache | tee gpurun_out/r03_cache_counters.txt
