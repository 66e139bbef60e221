cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_bvh_build.py -m gpu -x -q > $O/gputests6.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -30 $O/gputests6.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python scripts/bvh_build_bench.py r04 > $O/bvh_bench.log 2>&1
echo "bvh bench rc=$?"; cat $O/bvh_bench.log | cut -c1-900
