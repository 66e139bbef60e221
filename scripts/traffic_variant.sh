#!/bin/bash
# scripts/traffic_variant.sh "<variant names>" "<kernel symbol substring>" [bench args...] -- FETCH_SIZE / WRITE_SIZE of one kernel for library
# builds made by scripts/build_variant.sh ("base" = the in-tree library), then the bench lines of the same variants, alternating
# (inside gpurun; one --pmc counter per pass with --kernel-trace only, as the pool requires).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
names=$1; sym=$2; shift 2
for lib in $names; do
  L=polaris_amd/lib/libpolaris_hip.so; [ "$lib" != base ] && L=gpurun_in/variants/$lib.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/tv
    POLARIS_HIP_LIB=$L rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/tv" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timers "$@" > /dev/null 2>&1
    echo "$lib $c $(python3 scripts/pmc_sum.py gpurun_out/tv | grep -F "$sym" | sed 's/.*launches/launches/')"
  done
done
bash scripts/ab_variants.sh "$names $names $names" "$@"
