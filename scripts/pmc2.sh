#!/bin/bash
# scripts/pmc2.sh <outdir> "<counters>" <bench args...> -- one counter pass over one bench frame (inside gpurun)
export TMPDIR=/tmp
out=$1; shift; ctrs=$1; shift
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timers "$@" > /dev/null 2>&1
python3 - <<PY
import csv, collections, glob
f = glob.glob('gpurun_out/$out/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float)
seen=set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][-30:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); dur[k] += (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
for k, d in agg.items():
    if 'pol::' not in k or dur[k] < 0.5: continue
    print(k, 'ms=%.2f' % dur[k], ' '.join('%s=%.4g' % (c, v) for c, v in sorted(d.items())))
PY
