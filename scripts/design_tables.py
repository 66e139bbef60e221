#!/usr/bin/env python3
"""scripts/design_tables.py <tag> -- the markdown tables of DESIGN.md section 6 from profiles/<tag>_bench.json and <tag>_configs.json."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
d = json.loads(open(f"profiles/{tag}_bench.json").read().strip().splitlines()[-1])
print(f"headline: {d['value']:.0f} Mrays/s, {d['ms_per_frame']:.2f} ms per frame; cpu baseline {d['cpu_baseline']}")
print()
print("| kernel symbol (= `roofline_per_kernel` key of the bench line) | ms / frame | launches | GB/s on its own streams | `frac` (priced by) | `frac_reference` | HBM traffic (PMC) vs algorithmic, per launch | lanes live per VALU instr. |")
print("|---|---|---|---|---|---|---|---|")
for k, v in sorted(d["roofline_per_kernel"].items(), key=lambda kv: -kv[1]["ms_per_frame"]):
    tr = f"{v['traffic'] / 1e6:.0f} MB vs {v['algorithmic_bytes_per_launch'] / 1e6:.0f} MB" if v.get("traffic") else "—"
    lu = f"{v['lane_util']:.2f}" if v.get("lane_util") else "—"
    dom = " — **the `roofline` object**" if k == d["roofline"]["kernel"] else ""
    print(f"| `{k}`{dom} | {v['ms_per_frame']:.2f} | {v['launches']} | {v['achieved']:.0f} | {v['frac']:.3f} ({v['priced_by']}) | {v['frac_reference']:.3f} | {tr} | {lu} |")
print()
try:
    c = json.load(open(f"profiles/{tag}_configs.json"))["configs"]
    print("| config | scene | Mrays/s | ms/frame | dominant kernel, `frac` |")
    print("|---|---|---|---|---|")
    for k, v in c.items():
        if "error" in v:
            print(f"| {k} | {v.get('what', '')} | error | | {v['error'][-80:]} |")
        elif k == "C1":
            print(f"| C1 | {v['what']} ({v['cores']} threads) | {v['reference_order']['Mrays_per_s']} in the reference's schedule, {v['sample_parallel']['Mrays_per_s']} sample-parallel | {v['reference_order']['ms_per_frame']} / {v['sample_parallel']['ms_per_frame']} | — |")
        else:
            r = v.get("roofline") or {}
            print(f"| {k} | {v['what']} | {v['Mrays_per_s']:.0f} | {v['ms_per_frame']:.2f} | `{r.get('kernel', '')}` {r.get('frac', 0):.3f} |")
except FileNotFoundError:
    pass
