#!/bin/bash
# scripts/pmc_big.sh <tag> -- counter evidence for the big-scene configurations (run inside gpurun):
#   per config (C4 at 32 of 512 spp, C5, terrain): rocprofv3 kernel stats, one SQ pass, one cache pass (TA / TCP / TCC),
#   FETCH_SIZE and WRITE_SIZE passes.  Summaries land in gpurun_out/<tag>_big_<config>_*.{csv,txt,json}.
export TMPDIR=/tmp
tag=$1
R=$GRAFT_REPO_ROOT
cd $R
declare -A CFG
CFG[C4]="--scene material-ball --width 1920 --height 1080 --spp 32"
CFG[C5]="--scene instanced --width 2048 --height 2048 --spp 16"
CFG[terrain]="--scene terrain --width 1024 --height 1024 --spp 32"
for c in ${CONFIGS:-C4 C5 terrain}; do
  args="${CFG[$c]} $EXTRA_BENCH_ARGS"
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-counters $args > gpurun_out/${tag}_big_${c}_bench.json 2> gpurun_out/${tag}_big_${c}_bench.err || exit 1
  rm -rf gpurun_out/bigprof
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bigprof -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-counters --opt overlap=1 $args > /dev/null 2>&1 || exit 1
  cp $(ls gpurun_out/bigprof/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_big_${c}_kernel_stats_overlap1.csv
  for pass in sq cache fetch write; do
    case $pass in
      sq) ctr="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY";;
      cache) ctr="${CACHE_COUNTERS:-TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum}";;
      fetch) ctr="FETCH_SIZE";;
      write) ctr="WRITE_SIZE";;
    esac
    rm -rf gpurun_out/bigpmc_$pass
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/bigpmc_$pass -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timers $args > /dev/null 2> gpurun_out/${tag}_big_${c}_pmc_$pass.err || { echo "pass $pass failed for $c"; tail -3 gpurun_out/${tag}_big_${c}_pmc_$pass.err; continue; }
    python3 scripts/pmc_sum.py gpurun_out/bigpmc_$pass > gpurun_out/${tag}_big_${c}_pmc_$pass.txt
  done
  echo "== $c"; cut -c1-400 gpurun_out/${tag}_big_${c}_bench.json; head -8 gpurun_out/${tag}_big_${c}_kernel_stats_overlap1.csv; cat gpurun_out/${tag}_big_${c}_pmc_*.txt
done
