#!/bin/bash
# scripts/ab_raw.sh -- like ab.sh but prints ms_per_step / value and the timed-region kernel table
for spec in "$@"; do
  label="${spec%%:*}"; args="${spec#*:}"
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$label', 'ms/step=%.2f Mrays/s=%.0f' % (d['ms_per_step'], d['value']), 'iso:', ' '.join('%s=%.2f' % (n, v) for n, v in d.get('kernels_isolated_ms_per_frame',{}).items() if v>0.02))
"
done
