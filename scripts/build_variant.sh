#!/bin/bash
# scripts/build_variant.sh <name> [--patch <experiment> ...] [-Dmacro=value ...] -- A/B / instrumented build of the HIP library into
# gpurun_in/variants/<name>.so (picked up with POLARIS_HIP_LIB=gpurun_in/variants/<name>.so; gpurun_in/ is git-ignored but travels
# with gpurun, and nothing under polaris_amd/ ships an experiment).
#
# The product sources carry NO experiment code.  What an instrumented or experimental build needs is kept as a patch against them
# under polaris_amd/csrc/experiments/ (tests/test_abi.py checks that every patch still applies):
#     --patch profile_loops     + -DPOLARIS_PROFILE_LOOPS      wave-level loop counters of k_trace (scripts/loop_profile.sh)
#     --patch profile_prologue  + -DPOLARIS_PROFILE_PROLOGUE   s_memrealtime stamps around k_trace's LDS staging
#     --patch reorder           + -DPOLARIS_EXP_REORDER        the coherence-reorder experiment (scripts/wave_lines.py; closed: EXPERIMENTS.md)
#     --patch timing_inexact    + -DPOLARIS_TIMING_FMA_SLAB / -DPOLARIS_TIMING_CONTRACT_MT   TIMING-ONLY inexact arithmetic (profiles/r06_price_of_exactness.txt)
#     --patch tiny_lds_transposed  [-DPOLARIS_AB_B128]        the tiny mode's pair records transposed in LDS (closed: profiles/r06_tiny_lds_layout_ab.txt)
#     --patch trace_spill       (option trace_spill=1)         16 stack entries in LDS + a global overflow column: 8 workgroups per CU (closed: profiles/r06_occupancy_curve.txt)
# (the three combine: apply them in the order reorder, profile_loops, profile_prologue -- scripts/wave_lines.py needs the first two.)
# Only the tracer's translation unit is rebuilt, from a patched COPY of polaris_amd/csrc in a temporary directory; the device BVH
# builder's object is the in-tree one (run `make -C polaris_amd/csrc` first).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
patches=()
while [ "$1" = "--patch" ]; do patches+=("$2"); shift 2; done
OUT=$ROOT/gpurun_in/variants
mkdir -p "$OUT"
SRC=$(mktemp -d /tmp/polaris_variant.XXXXXX)
trap 'rm -rf "$SRC"' EXIT
cp "$ROOT"/polaris_amd/csrc/*.h "$ROOT"/polaris_amd/csrc/*.hip "$SRC"/
for p in "${patches[@]}"; do
  (cd "$SRC" && patch -s -p1 < "$ROOT/polaris_amd/csrc/experiments/$p.patch") || { echo "build_variant.sh: experiments/$p.patch does not apply" >&2; exit 1; }
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -Wall -Wno-unused-function \
  -I"$ROOT/include" -I"$SRC" "$@" -c "$SRC/polaris_hip.hip" -o "$OUT/$name.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$OUT/$name.o" "$ROOT/polaris_amd/lib/obj/bvh_build.o" -o "$OUT/$name.so"
rm -f "$OUT/$name.o"
echo "$OUT/$name.so"
