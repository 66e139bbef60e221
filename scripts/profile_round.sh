#!/bin/bash
# scripts/profile_round.sh <tag> -- the evidence files of one round (run inside gpurun; copy gpurun_out/<tag>_* to profiles/):
#   <tag>_bench.json                      the default bench.py line
#   <tag>_kernel_stats_overlap{1,4}.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 20 --warmup 5` (one batch at a time / default)
#   <tag>_bench_under_rocprof.json        the bench line of that profiled run
#   <tag>_traffic.json                    PMC FETCH_SIZE / WRITE_SIZE passes (scripts/traffic.sh)
#   <tag>_sq_counters.{txt,json}          one SQ pass (scripts/pmc.sh): VALU instructions per wave, lane utilisation, waiting share
export TMPDIR=/tmp
tag=$1
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
for ov in 1 4; do
  rm -rf gpurun_out/prof_$ov
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$ov -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-counters --opt overlap=$ov > gpurun_out/${tag}_bench_under_rocprof_overlap$ov.json 2>/dev/null
  f=$(ls gpurun_out/prof_$ov/*/*kernel_stats.csv | head -1)
  cp $f gpurun_out/${tag}_kernel_stats_overlap$ov.csv
done
bash scripts/traffic.sh > gpurun_out/${tag}_traffic.txt 2>&1
cp gpurun_out/traffic.json gpurun_out/${tag}_traffic.json
bash scripts/pmc.sh ${tag} > gpurun_out/${tag}_pmc.log 2>&1 || echo "profile_round.sh: the SQ counter pass failed, see gpurun_out/${tag}_pmc.log"
head -12 gpurun_out/${tag}_kernel_stats_overlap1.csv
cat gpurun_out/${tag}_bench.json | cut -c1-1500
