#!/bin/bash
# scripts/pmc_pass.sh <tag> <pass name> "<counters>" <bench args...> -- ONE rocprofv3 counter pass over one bench frame (inside
# gpurun; --pmc only with --kernel-trace, as the pool requires).  Writes gpurun_out/<tag>_<pass>_counters.txt (+ .json keyed by
# kernel symbol); rocprofv3's own output is kept in gpurun_out/<tag>_<pass>.err and a failed pass stops the script.
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
tag=$1; pass=$2; counters=$3; shift 3
dir=gpurun_out/pmc_${tag}_${pass}
rm -rf "$dir"
# shellcheck disable=SC2086
rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$dir" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timers "$@" \
	> "gpurun_out/${tag}_${pass}.err" 2>&1
rc=$?
if [ $rc -ne 0 ]; then echo "pmc_pass.sh: rocprofv3 pass '$pass' failed (exit $rc): see gpurun_out/${tag}_${pass}.err" >&2; tail -5 "gpurun_out/${tag}_${pass}.err" >&2; exit $rc; fi
python3 scripts/pmc_sum.py "$dir" --json "gpurun_out/${tag}_${pass}_counters.json" | tee "gpurun_out/${tag}_${pass}_counters.txt"
