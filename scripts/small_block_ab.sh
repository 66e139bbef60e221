cd $GRAFT_REPO_ROOT
for o in "-" "traversal=0" "trace_wgs_per_cu=2" "samples_per_batch=32" "samples_per_batch=128" "packet_primary=1" "shade_wave=0"; do
  opts=""; [ "$o" != "-" ] && opts="--opt $o"
  for rep in 1 2; do
  timeout -k 10 120 python bench.py --emulate-rank 3/8 --steps 20 --warmup 5 --no-cpu-baseline --no-live-counters $opts 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels_isolated_ms_per_frame']
        print('%-24s %8.3f ms/frame  ' % ('$o', d['ms_per_frame']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b > 0.03))
"
  done
done
