set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
hipcc --offload-arch=gfx950 -O3 tests/tools/gather_rate.hip -o /tmp/gather_rate 2>/dev/null && timeout -k 10 300 /tmp/gather_rate 1000 > gpurun_out/r04/gather_rate.txt 2>&1
echo "gather rc=$?"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputests.log 2>&1
echo "pytest rc=$?"
tail -5 gpurun_out/r04/gputests.log
timeout -k 10 300 python bench.py > gpurun_out/r04/bench1.json 2> gpurun_out/r04/bench1.err
echo "bench rc=$?"
cut -c1-600 gpurun_out/r04/bench1.json
