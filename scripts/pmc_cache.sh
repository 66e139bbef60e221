#!/bin/bash
# scripts/pmc_cache.sh <tag> <bench args...> -- one TA / TCP / TCC counter pass over one bench frame (inside gpurun)
tag=$1; shift
exec "$(dirname "$0")/pmc_pass.sh" "$tag" cache "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "$@"
