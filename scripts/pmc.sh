#!/bin/bash
# scripts/pmc.sh <outdir> <bench args...> -- one SQ counter pass over one bench frame (inside gpurun)
export TMPDIR=/tmp
out=$1; shift
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timers "$@" > /dev/null 2>&1
python3 scripts/pmc_report.py gpurun_out/$out
