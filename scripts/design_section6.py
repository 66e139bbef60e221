#!/usr/bin/env python3
"""scripts/design_section6.py [tag] -- rewrites DESIGN.md section 6 (the measured tables) from profiles/<tag>_bench.json, <tag>_configs.json,
<tag>_kernel_stats_overlap1.csv, <tag>_traffic.json, <tag>_sq_counters.json and <tag>_big_*_bench.json, so that the prose numbers and the
committed evidence cannot drift apart."""
import csv
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda name: os.path.join(ROOT, "profiles", f"{tag}_{name}")
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a = s.index('## 6. Measured')
b = s.index('## 7. Multi-GPU')
d = json.loads(open(P("bench.json")).read().strip().splitlines()[-1])
cfg = json.load(open(P("configs.json")))['configs']
traffic = json.load(open(P("traffic.json")))["kernels"]
sq = json.load(open(P("sq_counters.json")))["kernels"]
rows = []
for k, v in sorted(d['roofline_per_kernel'].items(), key=lambda kv: -kv[1]['ms_per_frame']):
    star = ' — **the `roofline` object**' if k == d['roofline']['kernel'] else ''
    t = traffic.get(k)
    tr = (t["hbm_read_bytes"] + t["hbm_write_bytes"]) / t["launches"] if t and t.get("launches") else None
    q = sq.get(k, {})
    lane = q.get("lane_util")
    iss = bench.issue_roofline(q["SQ_INSTS_VALU"] / q["launches"], v["launches"], v["ms_per_frame"], lane, 256) if q.get("SQ_INSTS_VALU") else None
    rows.append(f"| `{k}`{star} | {v['ms_per_frame']:.2f} | {v['launches']} | {v['frac']:.3f} ({v['priced_by']}) | {v['frac_reference']:.3f} | "
                f"{(tr / 1e6):.0f} MB vs {v['algorithmic_bytes_per_launch'] / 1e6:.0f} MB = {tr / v['algorithmic_bytes_per_launch']:.2f} × | "
                f"{('%.2f' % lane) if lane else '—'} | {('%.2f / %.2f' % (iss['frac'], iss['frac_useful_lanes'])) if iss else '—'} |")
crow = []
for key in ['C2', 'headline', 'headline-refbvh', 'C3', 'C4', 'C5', 'C5-16spp', 'terrain']:
    c = cfg[key]
    r = c.get('roofline') or {}
    what = c['what'].split(' (configs')[0].split(' (58,682')[0].split(' (1,060')[0].split(' (1.0 M')[0]
    crow.append(f"| {key} | {what} | {c['Mrays_per_s']:.0f} | {c['ms_per_frame']:.2f} | `{r.get('kernel', '').replace('pol::', '')}` {r.get('frac', 0):.3f} |")
c1 = cfg['C1']
import re
big = {"lane": [], "valu": [], "wait": [], "l2": []}
for c in ("C4", "C5", "terrain"):
    try:
        sqt = open(P(f"big_{c}_pmc_sq.txt")).read()
        cat = open(P(f"big_{c}_pmc_cache.txt")).read()
        l = [x for x in sqt.splitlines() if "k_trace<false" in x][0]
        big["lane"].append(re.search(r"lane_util=([\d.]+)", l).group(1)[:4])
        big["valu"].append(re.search(r"valu_busy=([\d.]+)", l).group(1)[:4])
        big["wait"].append(re.search(r"wait_any=([\d.]+)", l).group(1)[:4])
        big["l2"].append(re.search(r"l2_hit=([\d.]+)", [x for x in cat.splitlines() if "k_trace<false" in x][0]).group(1)[:4])
    except (OSError, IndexError, AttributeError):
        for k in big:
            big[k].append("n/a")
big = {k: " / ".join(v) for k, v in big.items()}
pops = "; ".join(f"{k} {cfg[k]['ray_counts']['shadow_rays'] / 1e6:.1f} M, {100 * cfg[k]['ray_counts']['occluded_fraction']:.0f} %" for k in ("headline", "C4", "C5-16spp", "terrain") if cfg[k].get("ray_counts"))
ks = list(csv.DictReader(open(P("kernel_stats_overlap1.csv"))))
top = [r for r in ks if 'k_trace<false' in r['Name']][0]
iso = sum(v for k, v in d['kernels_isolated_ms_per_frame'].items() if k != 'shade')
R = d['roofline']
Rt = traffic[R['kernel']]
Rtr = (Rt["hbm_read_bytes"] + Rt["hbm_write_bytes"]) / Rt["launches"]
new6 = f'''## 6. Measured (MI355X, round {int(tag[1:])}) — evidence under `profiles/{tag}_*`

(Earlier rounds: `EXPERIMENTS.md`.  Headline frame 20.2 → 11.3 ms over rounds 1-2, 11.5 → 10.3 in round 4; round 5 moved bytes, round 6 measured what is left
(§3.1, §8) and changed no kernel on the path: the numbers below are round 5's kernels re-measured on this round's build.)

`bench.py`, headline (layered Cornell box, 808 triangles, 512² × 128 spp, 5 bounces, RR from bounce 3; 163.5 M rays per frame):
**{d['value']:.0f} Mrays/s, {d['ms_per_frame']:.2f} ms per frame** (`{tag}_bench.json`; 10.3-10.5 ms across the round's leases; `config.devices` names the GPU it ran on, `config.ray_counts` the ray
population: {d['config']['ray_counts']['shadow_rays'] / 1e6:.1f} M shadow rays per frame, {100 * d['config']['ray_counts']['occluded_fraction']:.1f} % of them occluded).  Per kernel SYMBOL, one batch at a
time (HIP events of the extra frame `bench.py` traces after its timed region; `{tag}_kernel_stats_overlap1.csv` — rocprofv3
`--kernel-trace --stats` of the same command — agrees: `k_trace<false, 16, 2, true>` {float(top['AverageNs'])/1e3:.1f} µs average over {top['Calls']} calls against
{R['avg_launch_ms']*1e3:.1f} µs).  `frac` = HBM roofline on the smaller of the two prices (§3); traffic = FETCH_SIZE × 2 + WRITE_SIZE (`{tag}_traffic.json`);
issue = `roofline_issue`: share of the chip's vector-issue rate, and the same × live lanes (`{tag}_sq_counters.json`).  Since round 5
`bench.py` collects these counters itself at the end of every N = 1 run (three rocprofv3 child passes, ~10 s): the driver's own
bench line carries the traffic of the build it timed (`counters_source`); the committed files are the fallback:

| kernel symbol (= `roofline_per_kernel` key) | ms / frame | launches | `frac` (priced by) | `frac_reference` | HBM traffic (PMC) vs algorithmic, per launch | live lanes per VALU instr. | issue / × lanes |
|---|---|---|---|---|---|---|---|
''' + "\n".join(rows) + f'''

The dominant kernel is the closest-hit traversal of camera AND bounce rays: {R['frac']:.3f} of the HBM roof on the bytes its own streams
move (40 B per bounce ray, 28 per camera ray), {R['frac_reference']:.3f} on the reference's (60 / 44 B), measured traffic {Rtr/1e6:.0f} MB per launch =
{Rtr/R['algorithmic_bytes_per_launch']:.2f} × algorithmic (12-byte hit records stored lane by lane as rays end).  It is bound by vector-instruction issue, not HBM:
the issue column is the figure that says how well it runs; the HBM fraction says how few bytes the formulation needs.  Isolated kernel
times add up to {iso:.2f} ms per frame; two batches in flight bring the frame to {d['ms_per_frame']:.2f}.  (`k_generate` reads nothing: its rate is
the rate at which L2 / the Infinity Cache take stores.)

All configurations (`scripts/configs.py` → `{tag}_configs.json`; every one at its FULL frame size and sample count; three timed frames
each, which reads 1-3 % above `bench.py`'s twenty):

| config | scene | Mrays/s | ms/frame | dominant kernel, `frac` |
|---|---|---|---|---|
| C1 | sphere 256² × 16 spp on the CPU restatement (stand-in for the reference's OpenCL CPU device; {c1['cores']} threads) | {c1['reference_order']['Mrays_per_s']} in the reference's schedule, {c1['sample_parallel']['Mrays_per_s']} sample-parallel | {c1['reference_order']['ms_per_frame']} / {c1['sample_parallel']['ms_per_frame']} | — |
''' + "\n".join(crow) + f'''

Big scenes, counters re-collected on this round's build (`{tag}_big_{{C4,C5,terrain}}_pmc_{{sq,cache,fetch,write}}.txt`): `k_trace<false, 24, 0, false>`
live lanes {big['lane']}, VALU busy {big['valu']} of wave cycles, waves waiting {big['wait']}, L2 hit rate
{big['l2']} (C4 / C5 / terrain): the gather-bound picture of round 3, unchanged (§3.1; the residency curve: `{tag}_occupancy_curve.txt`).
Ray populations (`config.ray_counts` in `{tag}_configs.json`): shadow rays per frame and their occluded share — {pops}.

CPU baseline (oracle, sample-parallel OpenMP on the 16 CPUs the box grants): {d['cpu_baseline']['sample'].split(',')[1].strip()} of the same frame: {d['cpu_baseline']['value']:.1f} Mrays/s,
{d['cpu_baseline']['ms_per_frame_extrapolated']/1e3:.2f} s per 128-spp frame — baseline only.  The boundary hands over no per-frame host buffers (scene and camera are
uploaded once; a `Trace` call moves a few KB of seeds and ≈ 1 KB of counters), so there is no separate PCIe-inclusive rate to quote.

'''
open(p, 'w').write(s[:a] + new6 + s[b:])
print("DESIGN.md section 6 rewritten:", d['value'], d['ms_per_frame'], len(new6), "bytes")
