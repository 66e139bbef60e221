#!/bin/bash
# round 6, item 3 continued: the eight-workgroups-per-CU variant of the big-scene traversal (option trace_spill=1) against the all-LDS stack
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spill or big_scene_kernel_variants" 2>&1 | tail -5 || exit 1
declare -A CFG
CFG[C4]="--scene material-ball --width 1920 --height 1080 --spp 32"
CFG[C5]="--scene instanced --width 2048 --height 2048 --spp 16"
CFG[terrain]="--scene terrain --width 1024 --height 1024 --spp 32"
for c in terrain C4 C5; do
  echo "== $c: occupancy curve with the spill variant (overlap 1)"
  EXTRA_OPTS=trace_spill=1 PER_CU=4,5,6,7,8 timeout -k 10 400 python3 scripts/occupancy_curve.py $c 2> gpurun_out/r06_spill_${c}.err | tee gpurun_out/r06_spill_${c}_curve.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  spill per_cu', d['trace_wgs_per_cu'], 'frame', d['frame_ms_overlap1'], 'closest', d['closest_hit_ms_per_frame'], 'any', d['any_hit_ms_per_frame'], d['closest_symbol'])"
  echo "== $c: bench (default overlap), alternating"
  for i in 1 2 3; do
    for o in "" "--opt trace_spill=1" "--opt trace_spill=1 --opt trace_wgs_per_cu=6" "--opt trace_spill=1 --opt trace_wgs_per_cu=5"; do
      timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-counters ${CFG[$c]} $o 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['kernels_isolated_ms_per_frame']
        print('  %-48s %8.0f Mrays/s %8.3f ms  intersect=%.2f occlusion=%.2f shade=%.2f' % ('$o' or 'base', d['value'], d['ms_per_frame'], k['intersect'], k['occlusion'], k['shade']))"
    done
  done
done 2>&1 | tee gpurun_out/r06_spill_ab.txt
