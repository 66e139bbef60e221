# scripts/trace_kernels.sh -- per-launch durations of the shade kernels of one frame, in launch order (rocprofv3 --kernel-trace; inside gpurun).
# VARIANTS="name:bench args ..." picks library builds made by scripts/build_variant.sh ("base" = the in-tree library).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for mode in "${VARIANTS:-base:}"; do
  lib=${mode%%:*}; opt=${mode#*:}
  rm -rf gpurun_out/kt
  POLARIS_HIP_LIB=$([ "$lib" = base ] && echo polaris_amd/lib/libpolaris_hip.so || echo gpurun_in/variants/$lib.so) rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timers --opt overlap=1 $opt > /dev/null 2>&1
  echo "== $lib $opt"
  python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/kt/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_shade' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows)//2:]   # the timed frame (second half)
print(' '.join('%s:%d' % (r['Kernel_Name'].split('<')[0].replace('void pol::k_','') + ('W' if 'wave' in r['Kernel_Name'] else ''), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) // 1000) for r in rows))
PY
done
