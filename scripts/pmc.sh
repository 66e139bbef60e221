#!/bin/bash
# scripts/pmc.sh <tag> <bench args...> -- one SQ counter pass over one bench frame (inside gpurun):
# gpurun_out/<tag>_sq_counters.txt (readable) and .json (keyed by kernel symbol; bench.py reads the committed copy's lane_util)
tag=$1; shift
exec "$(dirname "$0")/pmc_pass.sh" "$tag" sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "$@"
