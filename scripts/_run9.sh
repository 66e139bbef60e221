cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 1000 python scripts/configs.py r04 > $O/configs.log 2>&1
echo "configs rc=$?"; cat $O/configs.log | cut -c1-200
timeout -k 10 600 python scripts/bvh_build_bench.py r04 > $O/bvh_bench_final.log 2>&1
echo "bvh rc=$?"; cut -c1-300 $O/bvh_bench_final.log
