#!/bin/bash
# scripts/emulate_ranks.sh -- on ONE GPU, the frame time of single ranks of an N-GPU run (bench.py --emulate-rank R/N): DESIGN.md 7
cd $GRAFT_REPO_ROOT
for rn in 0/2 1/2 0/4 3/4 0/8 3/8 7/8; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers --emulate-rank $rn 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$rn', 'ms/frame=%.2f' % d['ms_per_frame'], 'Mrays/s=%.0f' % d['value'])"
done
