cd $GRAFT_REPO_ROOT
for v in "" "--bvh device" "--bvh device --bvh-algorithm lbvh"; do
  for sc in "--scene terrain --width 1024 --height 1024 --spp 32" "--scene material-ball --width 1920 --height 1080 --spp 32" ""; do
    timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $sc $v 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$sc | $v |', round(d['ms_per_frame'],2), d['roofline']['kernel'], round(d['roofline']['ms_per_frame'],2), [ (k.split('::')[1][:22], round(x['ms_per_frame'],2)) for k,x in d['roofline_per_kernel'].items() if 'trace' in k])
    elif 'Error' in l or 'error' in l: print('$sc | $v |', l.strip()[-150:])"
  done
done
