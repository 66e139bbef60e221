cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_probes.py -m gpu -x -q -k "not full_sample and not other_configs and not soak and not timing" > $O/gputests4.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $O/gputests4.log
[ $rc -ne 0 ] && exit 1
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('default     %.0f Mrays/s %.3f ms ' % (d['value'], d['ms_per_frame']), d['kernels_isolated_ms_per_frame'])"; done | tee $O/bench4.txt
timeout -k 10 200 python bench.py --no-cpu-baseline --opt shade_split=1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('shade_split %.0f Mrays/s %.3f ms ' % (d['value'], d['ms_per_frame']), d['kernels_isolated_ms_per_frame'])" | tee -a $O/bench4.txt
timeout -k 10 300 bash scripts/traffic.sh > $O/traffic4.txt 2>&1; cp gpurun_out/traffic.json $O/traffic4.json; tail -14 $O/traffic4.txt
