#!/bin/bash
# scripts/traffic.sh -- HBM traffic of the bench kernels from PMC counters (inside gpurun).
# Two separate passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: they do not fit one pass),
# each with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
# Writes gpurun_out/traffic.json: per kernel, launches and bytes (FETCH_SIZE is in KiB and, on
# gfx950, counts 64 B per 128-B request of a wide coalesced stream: the read side is doubled).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R || exit 1
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timers $@"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 bench.py $ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 bench.py $ARGS > /dev/null 2>&1
python3 - <<'PY'
import csv, collections, glob, json
out = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0})
for d, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != key: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        out[k][key + "_KiB"] += float(r["Counter_Value"])
        if key == "FETCH_SIZE": out[k]["launches"] += 1
res = {}
for k, v in out.items():
    if "pol::" not in k: continue
    v["hbm_read_bytes"] = v["FETCH_SIZE_KiB"] * 1024 * 2      # gfx950: FETCH_SIZE reads 1/2 of a wide stream
    v["hbm_write_bytes"] = v["WRITE_SIZE_KiB"] * 1024
    v["hbm_bytes_per_launch"] = (v["hbm_read_bytes"] + v["hbm_write_bytes"]) / max(v["launches"], 1)
    res[k] = v
# the same numbers keyed by bench.py's kernel timers (what bench.py's roofline.traffic reads)
timers = {"generate": ("pol::k_generate",), "intersect_packet": ("pol::k_trace_packet<false",), "intersect": ("pol::k_trace<false",),
          "occlusion": ("pol::k_trace<true",), "shade": ("pol::k_shade<", "pol::k_shade_wave<")}
tm = {}
for t, prefixes in timers.items():
    ks = [v for k, v in res.items() if k.startswith(prefixes)]
    if ks:
        tm[t] = {"launches": sum(v["launches"] for v in ks), "hbm_bytes": sum(v["hbm_read_bytes"] + v["hbm_write_bytes"] for v in ks)}
        tm[t]["hbm_bytes_per_launch"] = tm[t]["hbm_bytes"] / max(tm[t]["launches"], 1)
json.dump({"command": "bench.py " + "--steps 1 --warmup 0", "timers": tm, "kernels": res,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B streaming request); gather traffic uncalibrated"},
          open("gpurun_out/traffic.json", "w"), indent=1)
for k, v in res.items():
    print("%-40s launches=%d read=%.2f GB write=%.2f GB" % (k[-40:], v["launches"], v["hbm_read_bytes"] / 1e9, v["hbm_write_bytes"] / 1e9))
PY
