#!/bin/bash
# scripts/ab.sh -- run bench.py variants back to back on the GPU box and print the kernel table.
# usage (inside gpurun): bash scripts/ab.sh "<label>:<bench args>" ...
for spec in "$@"; do
  label="${spec%%:*}"; args="${spec#*:}"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k={n:{'ms':v*d['steps']} for n,v in d.get('kernels_isolated_ms_per_frame',{}).items()}
print('$label', 'Mrays/s=%.0f ms/frame=%.2f' % (d['value'], d['ms_per_frame']), ' '.join('%s=%.1f' % (n, v['ms']/d['steps']) for n,v in k.items() if v['ms']>0.05))
"
done
