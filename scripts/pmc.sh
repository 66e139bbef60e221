#!/bin/bash
# scripts/pmc.sh <tag> <bench args...> -- one SQ counter pass over one bench frame (inside gpurun):
# gpurun_out/<tag>_sq_counters.txt (readable) and .json (keyed by kernel symbol; bench.py reads the committed copy's lane_util)
export TMPDIR=/tmp
tag=$1; shift
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timers "$@" > /dev/null 2>&1
python3 scripts/pmc_sum.py gpurun_out/pmc_sq --json gpurun_out/${tag}_sq_counters.json | tee gpurun_out/${tag}_sq_counters.txt
