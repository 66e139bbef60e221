#!/usr/bin/env python3
"""scripts/design_section6.py -- rewrites DESIGN.md section 6 (the measured tables) from profiles/r04_bench.json, r04_configs.json and
r04_kernel_stats_overlap1.csv, so that the prose numbers and the committed evidence cannot drift apart."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
a = s.index('## 6. Measured (MI355X, round 4)')
b = s.index('## 7. Multi-GPU')
d = json.loads(open(os.path.join(ROOT, 'profiles/r04_bench.json')).read().strip().splitlines()[-1])
cfg = json.load(open(os.path.join(ROOT, 'profiles/r04_configs.json')))['configs']
rows = []
for k, v in sorted(d['roofline_per_kernel'].items(), key=lambda kv: -kv[1]['ms_per_frame']):
    star = ' — **the `roofline` object**' if k == d['roofline']['kernel'] else ''
    rows.append(f"| `{k}`{star} | {v['ms_per_frame']:.2f} | {v['launches']} | {v['achieved']:.0f} | {v['frac']:.3f} ({v['priced_by']}) | {v['frac_reference']:.3f} | {v['traffic']/1e6:.0f} MB vs {v['algorithmic_bytes_per_launch']/1e6:.0f} MB | {('%.2f' % v['lane_util']) if v['lane_util'] else '—'} |")
crow = []
for key in ['C2', 'headline', 'headline-refbvh', 'C3', 'C4', 'C5', 'C5-16spp', 'terrain']:
    c = cfg[key]
    r = c.get('roofline') or {}
    crow.append(f"| {key} | {c['what']} | {c['Mrays_per_s']:.0f} | {c['ms_per_frame']:.2f} | `{r.get('kernel', '')}` {r.get('frac', 0):.3f} |")
c1 = cfg['C1']
ks = list(csv.DictReader(open(os.path.join(ROOT, 'profiles/r04_kernel_stats_overlap1.csv'))))
top = [r for r in ks if 'k_trace<false' in r['Name']][0]
iso = sum(v for k, v in d['kernels_isolated_ms_per_frame'].items() if k != 'shade')
R = d['roofline']
new6 = f'''## 6. Measured (MI355X, round 4) — evidence under `profiles/r04_*`

(Rounds 1-3: `EXPERIMENTS.md`.  20.2 ms → 11.3 ms per headline frame over rounds 1-2; round 3 moved nothing; round 4: 11.5 → 10.3.)

`bench.py`, headline (layered Cornell box, 808 triangles, 512² × 128 spp, 5 bounces, RR from bounce 3; 163.5 M rays per frame):
**{d['value']:.0f} Mrays/s, {d['ms_per_frame']:.2f} ms per frame** (`r04_bench.json`; 10.2 – 10.45 ms on the six leases that ran the final build — 15 970 / 10.24 under
`profile_round.sh` —; the round started at 11.2 – 11.6).  Per kernel SYMBOL, one batch at a time (HIP events of the extra frame `bench.py`
traces after its timed region; `r04_kernel_stats_overlap1.csv` — rocprofv3 `--kernel-trace --stats` of the same command — agrees:
`k_trace<false, 16, 2, true>` {float(top['AverageNs'])/1e3:.1f} µs average over {top['Calls']} calls against {R['avg_launch_ms']*1e3:.1f} µs); every kernel carries two prices (§3): the
reference's stream bytes (SURVEY §8d) and its own, `frac` uses the smaller:

| kernel symbol (= `roofline_per_kernel` key of the bench line) | ms / frame | launches | GB/s on its own streams | `frac` (priced by) | `frac_reference` | HBM traffic (PMC) vs algorithmic, per launch | lanes live per VALU instr. |
|---|---|---|---|---|---|---|---|
''' + "\n".join(rows) + f'''

The dominant kernel is the closest-hit traversal, now of camera AND bounce rays (10 launches per frame; the wave-packet kernel
is not used where the triangle records are in LDS): {R['frac']:.3f} of the HBM roof on the bytes its own streams move (48 B per bounce
ray, 32 B per camera ray: their common origin is one record), {R['frac_reference']:.3f} on the reference's (60 / 44 B); round 3: 0.143 on the reference's
bytes for the bounce rays alone.  Measured traffic {R['traffic']/1e6:.0f} MB per launch = {R['traffic_over_algorithmic']:.2f} × algorithmic (the 16-byte hit records that lanes
store one by one as their rays end: partial sectors).  The kernel is bound by vector instruction issue at {R['lane_util']:.2f} live lanes
(§3.1), not by HBM — `frac` went DOWN with the camera-origin change (16 B per camera ray less to move, same time) while the
frame got 2 % faster: for this kernel the fraction measures how few bytes the formulation needs, not how well it runs.
Isolated kernel times add up to {iso:.2f} ms per frame; two batches in flight bring the frame to {d['ms_per_frame']:.2f}.
(`k_generate` reads nothing and writes two streams: its rate is the rate at which stores are taken by L2 / the Infinity Cache — a
write-only kernel retires before its lines are in HBM — so its `frac` says "as fast as stores go", not an HBM measurement.)

All configurations (`scripts/configs.py` → `r04_configs.json`; every one at its FULL frame size and sample count; three timed frames
each, which reads 1-3 % above `bench.py`'s twenty):

| config | scene | Mrays/s | ms/frame | dominant kernel, `frac` |
|---|---|---|---|---|
| C1 | {c1['what']} ({c1['cores']} threads) | {c1['reference_order']['Mrays_per_s']} in the reference's schedule, {c1['sample_parallel']['Mrays_per_s']} sample-parallel | {c1['reference_order']['ms_per_frame']} / {c1['sample_parallel']['ms_per_frame']} | — |
''' + "\n".join(crow) + f'''

Against round 3: headline 11.53 → {cfg['headline']['ms_per_frame']:.2f} ms in this table's three-frame timing, C2 6.34 → {cfg['C2']['ms_per_frame']:.2f}, C3 85.8 → {cfg['C3']['ms_per_frame']:.1f}
(the tiny-mode work of §3.1); C5 4 843 Mrays/s at 16 spp → {cfg['C5-16spp']['Mrays_per_s']:.0f} ({cfg['C5']['Mrays_per_s']:.0f} at its full 1 024 spp: the instance handling); C4 361 → {cfg['C4']['ms_per_frame']:.0f} ms and the
terrain 22.3 → {cfg['terrain']['ms_per_frame']:.1f} (gather-bound, §3.1: the ray set-up's cheaper reciprocal, one stack write less per ray and the
12-byte records are all they got).

CPU baseline (oracle, sample-parallel OpenMP on the 16 CPUs the box grants): {d['cpu_baseline']['sample'].split(',')[1].strip()} of the same frame: {d['cpu_baseline']['value']:.1f} Mrays/s,
{d['cpu_baseline']['ms_per_frame_extrapolated']/1e3:.2f} s per 128-spp frame — baseline only.  The boundary hands over no per-frame host buffers (scene and camera are
uploaded once; a `Trace` call moves a few KB of seeds and ≈ 1 KB of counters), so there is no separate PCIe-inclusive rate to quote.

'''
open(p, 'w').write(s[:a] + new6 + s[b:])
print("DESIGN.md section 6 rewritten:", d['value'], d['ms_per_frame'])
