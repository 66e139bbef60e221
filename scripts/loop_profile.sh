#!/bin/bash
# scripts/loop_profile.sh [bench args...] -- where the traversal kernels' iterations go (inside gpurun; build the variant first, on the
# build machine: scripts/build_variant.sh prof --patch profile_loops -DPOLARIS_PROFILE_LOOPS).  Prints, for the closest-hit and the any-hit kernel of one
# frame: outer iterations, node steps and triangle rounds per ray, the lanes live in each, and the wave-level iteration counts
# (kernels.h PROF, polaris_hip.hip).  Evidence of round 4: profiles/r04_loop_profile_*.txt.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
[ -f gpurun_in/variants/prof.so ] || { echo "build gpurun_in/variants/prof.so first: scripts/build_variant.sh prof --patch profile_loops -DPOLARIS_PROFILE_LOOPS"; exit 1; }
POLARIS_DEBUG=1 POLARIS_HIP_LIB=gpurun_in/variants/prof.so timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-live-counters "$@" 2>&1 | grep "per ray" | tail -2
