#!/bin/bash
# scripts/ab_wide.sh [tag] -- A/B of the four-wide tree (option wide) on the gather-bound scenes, one bench.py run each (inside gpurun).
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
tag=${1:-wide_ab}
out=gpurun_out/${tag}.txt
: > "$out"
run() { # name, args...
	name=$1; shift
	for w in 0 1; do
		line=$(timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --opt wide=$w "$@" 2>gpurun_out/${tag}_err.txt | grep '^{') || { echo "$name wide=$w FAILED" >> "$out"; tail -5 gpurun_out/${tag}_err.txt >> "$out"; return 1; }
		python3 - "$name" "$w" "$line" >> "$out" <<'PY'
import json, sys
name, w, line = sys.argv[1:4]
d = json.loads(line)
k = d.get("kernels_isolated_ms_per_frame", {})
print(f"{name:14s} wide={w}  {d['ms_per_frame']:8.2f} ms/frame  {d['value']:8.0f} Mrays/s   closest {k.get('intersect', 0):7.2f}  any-hit {k.get('occlusion', 0):7.2f}  shade {k.get('shade', 0):7.2f}  [{d['roofline']['kernel']}]")
PY
		tail -1 "$out"
	done
}
run terrain --scene terrain --width 1024 --height 1024 --spp 32 &&
run C4-32spp --scene material-ball --width 1920 --height 1080 --spp 32 &&
run C5-16spp --scene instanced --width 2048 --height 2048 --spp 16 &&
run headline
