# scripts/traffic_variant.sh -- FETCH_SIZE / WRITE_SIZE of the any-hit kernel and the bench line for library builds made by scripts/build_variant.sh (inside gpurun)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for lib in base lateacc; do
  L=polaris_amd/lib/libpolaris_hip.so; [ $lib != base ] && L=polaris_amd/lib/exp/$lib.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/tv
    POLARIS_HIP_LIB=$L rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timers > /dev/null 2>&1
    echo "$lib $c $(python3 scripts/pmc_sum.py gpurun_out/tv | grep 'k_trace<true' | sed 's/.*launches/launches/')"
  done
done
bash scripts/ab_variants.sh "base lateacc base lateacc"
