"""scripts/pmc_sum.py <rocprofv3 output dir> [--json out.json] -- per kernel symbol: launches and the sum of every collected
counter, plus the derived SQ ratios when their counters are present.  --json: the same keyed by symbol (what bench.py's
`lane_util` reads: profiles/r*_sq_counters.json)."""
import collections
import csv
import glob
import json
import sys

found = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
if not found:
    sys.exit(f"pmc_sum.py: no *counter_collection.csv under {sys.argv[1]} -- the rocprofv3 pass produced nothing (did it fail?)")
f = found[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[k].add(r["Dispatch_Id"])
out = {}
for k, d in sorted(agg.items()):
    if "pol::" not in k:
        continue
    out[k] = {"launches": len(launches[k]), **{c: v for c, v in d.items()}}
    if d.get("SQ_WAVES") and d.get("SQ_ACTIVE_INST_VALU") and d.get("SQ_WAVE_CYCLES"):
        out[k].update({"valu_per_wave": d["SQ_INSTS_VALU"] / d["SQ_WAVES"], "lane_util": d["SQ_THREAD_CYCLES_VALU"] / (64 * d["SQ_ACTIVE_INST_VALU"]),
                       "valu_busy": d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"], "wait_any": d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]})
    line = "%-44s launches=%d" % (k[-44:], len(launches[k]))
    for c, v in sorted(d.items()):
        line += " %s=%.4g" % (c, v)
    if d.get("SQ_WAVES") and d.get("SQ_ACTIVE_INST_VALU") and d.get("SQ_WAVE_CYCLES"):
        line += " | valu/wave=%.0f lane_util=%.3f valu_busy=%.3f wait_any=%.3f wait_inst=%.3f" % (
            d["SQ_INSTS_VALU"] / d["SQ_WAVES"], d["SQ_THREAD_CYCLES_VALU"] / (64 * d["SQ_ACTIVE_INST_VALU"]),
            d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"], d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"])
    if d.get("TCC_HIT_sum") is not None and d.get("TCC_MISS_sum") is not None and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
        line += " | l2_hit=%.3f" % (d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]))
    print(line)
if "--json" in sys.argv:
    json.dump({"source": "rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 1 --warmup 0 (scripts/pmc.sh)", "kernels": out},
              open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
