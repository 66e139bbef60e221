#!/bin/bash
# scripts/pmc_lds.sh <tag> <bench args...> -- one LDS counter pass over one bench frame (inside gpurun): bank conflicts per kernel symbol
export TMPDIR=/tmp
tag=$1; shift
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_lds
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_lds -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timers "$@" > /dev/null 2> gpurun_out/${tag}_lds.err
python3 scripts/pmc_sum.py gpurun_out/pmc_lds | tee gpurun_out/${tag}_lds_counters.txt
