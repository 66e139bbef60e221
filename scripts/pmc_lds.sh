#!/bin/bash
# scripts/pmc_lds.sh <tag> <bench args...> -- one LDS counter pass over one bench frame (inside gpurun): bank conflicts per kernel symbol
tag=$1; shift
exec "$(dirname "$0")/pmc_pass.sh" "$tag" lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT" "$@"
