#!/bin/bash
# scripts/abx.sh -- like ab.sh, with a library per variant: "<label>|<lib or ->|<bench args>"
for spec in "$@"; do
  IFS='|' read -r label lib args <<< "$spec"
  if [ "$lib" != "-" ]; then export POLARIS_HIP_LIB=$GRAFT_REPO_ROOT/polaris_amd/lib/exp/$lib.so; else unset POLARIS_HIP_LIB; fi
  POLARIS_DEBUG=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args 2>gpurun_out/abx_$label.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d.get('kernels_isolated_ms_per_frame',{})
print('$label', 'Mrays/s=%.0f ms/frame=%.2f' % (d['value'], d['ms_per_frame']), ' '.join('%s=%.2f' % (n, v) for n,v in k.items() if v>0.05))
"
  grep -h "polaris\]\|\[trace" gpurun_out/abx_$label.err | sort | uniq -c | head -24
done
