#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { # name lib args
  local L=gpurun_in/variants/$2.so; shift; shift
  POLARIS_HIP_LIB=$L timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-live-counters "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels_isolated_ms_per_frame']
        print('%-10s %8.0f Mrays/s %7.3f ms  ' % ('$L'.split('/')[-1][:-3], d['value'], d['ms_per_frame']) + ' '.join('%s=%.2f' % (n, k[n]) for n in ('intersect', 'occlusion', 'shade')))"
}
for i in 1 2 3; do for v in base swz swz128; do run $v $v; done; done
echo "== sphere (C2)"; for i in 1 2; do for v in base swz swz128; do run $v $v --scene sphere; done; done
echo "== cubes-like multi-instance tiny scene (general tiny variant)"; for v in base swz swz128; do run $v $v --scene cubes --steps 10; done
export TMPDIR=/tmp
for v in base swz swz128; do
  rm -rf gpurun_out/swz_pmc
  POLARIS_HIP_LIB=gpurun_in/variants/$v.so rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/swz_pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timers --no-live-counters > /dev/null 2>&1 || { echo "pmc failed $v"; continue; }
  echo "== $v"; python3 scripts/pmc_sum.py gpurun_out/swz_pmc | grep "k_trace" | cut -c1-300
done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiny_scene or golden or exact_and_batched" 2>&1 | tail -3
