#!/bin/bash
# scripts/ab_opt.sh tag "optA" "optB" [bench args...] -- A/B of two tracer option sets (comma separated key=value, "-" = none) on bench.py (inside gpurun);
# three alternating runs each, isolated kernel times of bench.py's extra frame beside the frame time.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
tag=$1; a=$2; b=$3; shift 3
out=gpurun_out/${tag}.txt
: > "$out"
for rep in 1 2 3; do
	for o in "$a" "$b"; do
		opts=""
		if [ "$o" != "-" ]; then for kv in ${o//,/ }; do opts="$opts --opt $kv"; done; fi
		line=$(timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-live-counters $opts "$@" 2>gpurun_out/${tag}_err.txt | grep '^{') || { echo "[$o] FAILED" >> "$out"; tail -5 gpurun_out/${tag}_err.txt >> "$out"; exit 1; }
		python3 - "$o" "$line" >> "$out" <<'PY'
import json, sys
o, line = sys.argv[1:3]
d = json.loads(line)
k = d.get("kernels_isolated_ms_per_frame", {})
print(f"[{o:24s}] {d['ms_per_frame']:8.3f} ms/frame {d['value']:8.0f} Mrays/s | packet {k.get('intersect_packet', 0):5.2f} closest {k.get('intersect', 0):6.3f} any-hit {k.get('occlusion', 0):6.3f} shade {k.get('shade', 0):6.3f}")
PY
		tail -1 "$out"
	done
done
