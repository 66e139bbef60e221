#!/bin/bash
# scripts/ab_variants.sh "<variant names>" "<bench args>" [--opt k=v ...] -- A/B of library builds made by scripts/build_variant.sh
# (gpurun_in/variants/<name>.so; "base" = the in-tree library) on one bench configuration; prints Mrays/s, ms/frame and the
# isolated kernel times per variant.  Run inside gpurun.
cd $GRAFT_REPO_ROOT
names=$1; shift
for n in $names; do
  lib=gpurun_in/variants/$n.so
  [ "$n" = base ] && lib=polaris_amd/lib/libpolaris_hip.so
  POLARIS_HIP_LIB=$lib timeout -k 10 240 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-counters "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels_isolated_ms_per_frame']
        print('%-10s %7.0f Mrays/s %8.2f ms  ' % ('$n', d['value'], d['ms_per_frame']) + ' '.join('%s=%.2f' % (a, b) for a, b in k.items() if b > 0.3))
"
done
