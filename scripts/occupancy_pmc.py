#!/usr/bin/env python3
"""scripts/occupancy_pmc.py <rocprofv3 output dir> <CUs> -- counters of scripts/occupancy_curve.py --pmc, per traversal kernel symbol and
per SETTING: the persistent grid of k_trace is CUs x trace_wgs_per_cu workgroups of 256 threads, so a dispatch's grid size names
the setting it ran under (launches of a sparse late bounce have fewer chunks than the grid and are listed under `small`)."""
import collections
import csv
import glob
import json
import sys

found = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
if not found:
    sys.exit(f"no *counter_collection.csv under {sys.argv[1]}")
cus = int(sys.argv[2]) if len(sys.argv) > 2 else 256
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for r in csv.DictReader(open(found[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "k_trace<" not in k:
        continue
    grid = int(r["Grid_Size"])
    per_cu = grid // (cus * 256) if grid % (cus * 256) == 0 else 0
    key = (k, per_cu if 1 <= per_cu <= 8 else 0)
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[key].add(r["Dispatch_Id"])
for (k, per_cu), d in sorted(agg.items()):
    line = {"kernel": k, "trace_wgs_per_cu": per_cu or "small", "launches": len(launches[(k, per_cu)])}
    line.update({c: v for c, v in sorted(d.items())})
    if d.get("SQ_WAVE_CYCLES"):
        line["wait_any"] = round(d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"], 3)
        line["valu_busy"] = round(d.get("SQ_ACTIVE_INST_VALU", 0.0) / d["SQ_WAVE_CYCLES"], 3)
    if d.get("SQ_ACTIVE_INST_VALU"):
        line["lane_util"] = round(d.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64 * d["SQ_ACTIVE_INST_VALU"]), 3)
    if d.get("GRBM_GUI_ACTIVE") and d.get("TA_TA_BUSY_sum") is not None:
        # TA_TA_BUSY_sum adds the busy cycles of every CU's TA; GRBM_GUI_ACTIVE adds the active cycles of the 8 XCDs
        line["ta_busy"] = round((d["TA_TA_BUSY_sum"] / cus) / (d["GRBM_GUI_ACTIVE"] / 8.0), 3)
    if d.get("TCC_HIT_sum") is not None and (d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0)) > 0:
        line["l2_hit"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 3)
    print(json.dumps(line))
