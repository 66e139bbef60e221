#!/bin/bash
# round 6, item 4: what bit-exactness costs the dominant kernel -- TIMING-ONLY builds (results invalid), headline workload, alternating runs
cd $GRAFT_REPO_ROOT
run() { # name lib
  local L=polaris_amd/lib/libpolaris_hip.so; [ "$2" != base ] && L=gpurun_in/variants/$2.so
  POLARIS_HIP_LIB=$L timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-live-counters 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels_isolated_ms_per_frame']; r = d['roofline_per_kernel']
        t = [e for s, e in r.items() if s.startswith('pol::k_trace<false')][0]
        print('%-12s %8.0f Mrays/s %7.3f ms  closest-hit %.1f us/launch  ' % ('$1', d['value'], d['ms_per_frame'], t['avg_launch_ms'] * 1e3) + ' '.join('%s=%.2f' % (n, k[n]) for n in ('intersect', 'occlusion', 'shade_first', 'shade_sort', 'shade_wave', 'shade')))"
}
for i in 1 2 3; do
  for v in base fma_slab contract_mt fma_mt fastdiv all_fast; do run $v $v; done
done | tee gpurun_out/r06_exact_times.txt
# VALU instructions per wave of the dominant kernel, per variant (one SQ pass each)
export TMPDIR=/tmp
for v in base fma_slab contract_mt fma_mt all_fast; do
  L=polaris_amd/lib/libpolaris_hip.so; [ "$v" != base ] && L=gpurun_in/variants/$v.so
  rm -rf gpurun_out/exact_pmc
  POLARIS_HIP_LIB=$L rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/exact_pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timers --no-live-counters > /dev/null 2>&1 || { echo "pmc failed $v"; continue; }
  echo "== $v"; python3 scripts/pmc_sum.py gpurun_out/exact_pmc | grep "k_trace\|k_shade"
done | tee gpurun_out/r06_exact_valu.txt
rm -rf gpurun_out/exact_pmc
python -m pytest tests/test_gpu_bench.py -x -q -m gpu -k "masked or setup" 2>&1 | tail -5
