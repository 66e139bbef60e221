#!/bin/bash
# round 6, item 3: occupancy curve of the big-scene traversal kernels (scripts/occupancy_curve.py)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in terrain C4 C5; do
  echo "== $c times" | tee -a gpurun_out/r06_occ_progress.txt
  timeout -k 10 500 python3 scripts/occupancy_curve.py $c > gpurun_out/r06_occ_${c}_times.jsonl 2> gpurun_out/r06_occ_${c}_times.err || { echo "times failed $c"; tail -5 gpurun_out/r06_occ_${c}_times.err; exit 1; }
  cat gpurun_out/r06_occ_${c}_times.jsonl
  for pass in sq cache; do
    case $pass in
      sq) ctr="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY";;
      cache) ctr="TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum";;
    esac
    rm -rf gpurun_out/occpmc_$pass
    echo "== $c pmc $pass" | tee -a gpurun_out/r06_occ_progress.txt
    timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/occpmc_$pass -- python3 scripts/occupancy_curve.py $c --pmc > gpurun_out/r06_occ_${c}_pmc_$pass.log 2>&1 || { echo "pmc $pass failed $c"; tail -5 gpurun_out/r06_occ_${c}_pmc_$pass.log; exit 1; }
    python3 scripts/occupancy_pmc.py gpurun_out/occpmc_$pass 256 > gpurun_out/r06_occ_${c}_pmc_$pass.jsonl
    cat gpurun_out/r06_occ_${c}_pmc_$pass.jsonl
    rm -rf gpurun_out/occpmc_$pass
  done
done
