#!/usr/bin/env python3
"""scripts/wave_lines.py -- the go / no-go measurement for a coherence reorder of the bounce rays on the scenes whose tree does
not fit LDS (run inside gpurun; build the variant first on the build machine:
    scripts/build_variant.sh reorder --patch reorder --patch profile_loops -DPOLARIS_PROFILE_LOOPS -DPOLARIS_EXP_REORDER).

For terrain / C4 (material-ball) / C5 (instanced), in the kernel's REAL wave assignment (persistent waves, lane refill): distinct
128-byte lines per wave-level node step and per triangle round, lines per ray, and the time of the k_trace launches themselves,
with the bounce rays dealt (a) as k_shade emits them today, and in the orders the experiment build can impose between the shade
step and the launch (polaris_hip.hip, POLARIS_EXP_REORDER): identity through the indirection (the control: what the indirection
itself costs), direction octant + Morton code of the origin cell within 256-slot and 1 024-slot windows and over the whole batch,
Morton code alone (C5: stands for "the instance the ray starts in"), 16 x 16 pixel tiles.  Results are unchanged bit for bit by
any order (hits go to the ray's own slot); the script checks the ray counters and the accumulator against the plain run.

    python scripts/wave_lines.py r05        # -> gpurun_out/r05_wave_lines.txt (+ .json); the builder copies them to profiles/
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = [("terrain", "terrain", 1024, 1024, 32), ("C4", "material-ball", 1920, 1080, 32), ("C5", "instanced", 2048, 2048, 16)]
if os.environ.get("WAVE_LINES_SCENES"):   # e.g. "headline:cornell:512:512:128"
    SCENES = [(a, b, int(c), int(d), int(e)) for a, b, c, d, e in (x.split(":") for x in os.environ["WAVE_LINES_SCENES"].split(","))]
ORDERS = [  # (label, reorder, window, shadow rays too[, Morton bits kept: 30 = 10 per axis])
    ("a: as emitted today (no indirection)", 0, 256, 0),
    ("control: identity through the indirection", 1, 256, 0),
    ("b: octant, then Morton(origin), 256-slot chunk", 2, 256, 0),
    ("b': Morton(origin), then octant, 256-slot chunk", 3, 256, 0),
    ("c: octant, then Morton(origin), 1024-slot window", 2, 1024, 0),
    ("c': Morton(origin), then octant, 1024-slot window", 3, 1024, 0),
    ("d: Morton(origin) alone (~ first instance entered), 1024-slot window", 4, 1024, 0),
    ("e: octant, then Morton(origin), whole batch (upper bound)", 2, 0, 0),
    ("e': Morton(origin), then octant, whole batch (upper bound)", 3, 0, 0),
    ("f: 16x16 pixel tiles, whole batch", 5, 0, 0),
    ("g: octant alone, 1024-slot window", 6, 1024, 0),
    ("c + shadow rays: octant, Morton, 1024-slot window, any-hit launches too", 2, 1024, 1),
    ("h: whole batch, BINS of Morton(origin) top 9 bits (8^3 cells) x octant", 3, 0, 0, 9),
    ("h: whole batch, bins of Morton top 12 bits (16^3 cells) x octant", 3, 0, 0, 12),
    ("h: whole batch, bins of Morton top 15 bits (32^3 cells) x octant", 3, 0, 0, 15),
    ("h: whole batch, bins of Morton top 18 bits (64^3 cells) x octant", 3, 0, 0, 18),
    ("h + shadow rays: bins of Morton top 15 bits x octant, any-hit launches too", 3, 0, 1, 15),
    ("i: all samples of a pixel side by side (pixel-major), whole batch", 7, 0, 0),
    ("i + shadow rays: pixel-major, any-hit launches too", 7, 0, 1),
]
if os.environ.get("WAVE_LINES_ORDERS"):   # comma separated first letters, e.g. "a,control,i"
    keep = tuple(os.environ["WAVE_LINES_ORDERS"].split(","))
    ORDERS = [o for o in ORDERS if o[0].startswith(keep)]


def child(tag):
    sys.path.insert(0, ROOT)
    import numpy as np

    from polaris_amd import ctypes_api as T
    from polaris_amd import scenes
    from polaris_amd.tracer import ChangeType, HipTracer, UpdateMode

    def mark(s):
        os.write(2, ("### " + s + "\n").encode())

    for key, scene, W, H, spp in SCENES:
        sc = scenes.SCENES[scene](W / H)
        B = 5
        seeds = scenes.make_seeds(spp, B)
        tr = HipTracer("exp", 0)
        tr.Init()
        tr.set_option("overlap", 1)
        tr.set_option("time_kernels", 1)
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.FrameDimensions, (W, H))
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.SceneData, sc)
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.CameraData, sc)
        req = T.BlockRequest()
        req.frame_w, req.frame_h, req.block_x, req.block_y, req.block_w, req.block_h = W, H, 0, 0, W, H
        req.samples_per_pixel, req.num_bounces, req.min_bounces_for_rr = spp, B, 3
        req.exposure, req.seed, req.accumulated_samples = 1.2, 0, 0
        base = None
        for label, mode, win, anyhit, *rest in ORDERS:
            tr.set_option("reorder_bits", rest[0] if rest else 30)
            tr.set_option("reorder", mode)
            tr.set_option("reorder_win", win)
            tr.set_option("reorder_any", anyhit)
            mark(f"warm {key}")
            tr.Trace(req, seeds)
            for name in ("intersect", "occlusion"):
                tr.kernel_ms(name)
            mark(f"run {key} | {label}")
            t0 = __import__("time").perf_counter()
            tr.Trace(req, seeds)
            wall = (__import__("time").perf_counter() - t0) * 1e3
            st = tr.last_trace_stats
            acc = tr.read_accumulator(0)
            sig = (int(st.primary_rays), int(st.indirect_rays), int(st.occlusion_rays), int(st.unoccluded), acc.view(np.uint32).sum(dtype=np.uint64).item())
            if base is None:
                base = sig
            ims, ins = tr.kernel_ms("intersect")
            oms, ons = tr.kernel_ms("occlusion")
            mark(f"done {key} | {label} | intersect_ms={ims:.3f} launches={ins} occlusion_ms={oms:.3f} wall_ms={wall:.2f} same={sig == base} rays={sig[0] + sig[1] + sig[2]} tris={sc.num_triangles} frame={W}x{H}x{spp}")
        tr.Close()


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "--child":
        return child(sys.argv[1])
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    env = dict(os.environ, POLARIS_DEBUG="1", POLARIS_HIP_LIB=os.path.join(ROOT, "gpurun_in/variants/reorder.so"))
    p = subprocess.run([sys.executable, os.path.abspath(__file__), tag, "--child"], env=env, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    log = p.stderr
    open(os.path.join(ROOT, "gpurun_out", f"{tag}_wave_lines_raw.log"), "w").write(log)
    if p.returncode != 0:
        print(log[-3000:])
        raise SystemExit(p.returncode)
    rows, cur = [], None
    for line in log.splitlines():
        if line.startswith("### run "):
            cur = {"scene": line[8:].split(" | ")[0], "order": line.split(" | ", 1)[1]}
        elif line.startswith("### warm"):
            cur = None
        elif line.startswith("### done ") and cur is not None:
            for kv in line.split(" | ")[2].split():
                k, v = kv.split("=")
                cur[k] = v
            rows.append(cur)
            cur = None
        elif cur is not None and line.startswith("[polaris] "):
            m = re.match(r"\[polaris\] (closest hit|any hit) lines: per wave-level node step ([\d.]+) distinct lines for ([\d.]+) live lanes; per triangle round ([\d.]+) for ([\d.]+); "
                         r"per ray ([\d.]+) node lines \+ ([\d.]+) triangle lines; instance entries per ray ([\d.]+), distinct instance records per entering lane ([\d.]+)", line)
            if m:
                k = "ch" if m.group(1) == "closest hit" else "ah"
                cur[k] = dict(zip(("node_lines_per_step", "node_lanes", "tri_lines_per_round", "tri_lanes", "node_lines_per_ray", "tri_lines_per_ray", "inst_entries_per_ray", "inst_recs_per_entry"),
                                  (float(g) for g in m.groups()[1:])))
            m = re.match(r"\[polaris\] (closest hit|any hit): (\d+) rays; per ray: ([\d.]+) outer iterations, ([\d.]+) node steps, ([\d.]+) triangle rounds", line)
            if m:
                k = "ch_steps" if m.group(1) == "closest hit" else "ah_steps"
                cur[k] = {"rays": int(m.group(2)), "node_steps_per_ray": float(m.group(4)), "tri_rounds_per_ray": float(m.group(5))}
            m = re.match(r"\[polaris\] reorder .*alone ([\d.]+) ms closest hit, ([\d.]+) ms any hit", line)
            if m:
                cur["k_trace_bounce_ms"], cur["k_trace_shadow_ms"] = float(m.group(1)), float(m.group(2))
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", f"{tag}_wave_lines.json"), "w"), indent=1)
    out = [f"# {tag}: distinct 128-byte lines per wave-level step of k_trace (closest hit: camera + bounce rays of one frame) under imposed ray orders",
           "# scripts/wave_lines.py, experiment build -DPOLARIS_PROFILE_LOOPS -DPOLARIS_EXP_REORDER, overlap=1; every order leaves the result bit-identical (column `same`)",
           "# lines/ray = distinct lines summed over the wave-level steps / rays: what the CU's gather path is charged per ray; k_trace ms = the closest-hit launches of the BOUNCE rays alone",
           "# (events around the launch; the camera launch and the key + sort pass are outside), shadow ms = the any-hit launches", ""]
    hdr = f"{'scene':8} {'order':74} {'node lines/step':>15} {'lanes':>6} {'tri lines/round':>15} {'lanes':>6} {'lines/ray':>10} {'vs a':>6} {'k_trace ms':>10} {'vs ctl':>7} {'shadow ms':>9} {'same':>5}"
    base_lines, ctl_ms = {}, {}
    for r in rows:
        if "ch" not in r:
            continue
        lpr = r["ch"]["node_lines_per_ray"] + r["ch"]["tri_lines_per_ray"]
        if r["order"].startswith("a:"):
            base_lines[r["scene"]] = lpr
            out.append(hdr)
        if r["order"].startswith("control"):
            ctl_ms[r["scene"]] = r.get("k_trace_bounce_ms", 0.0)
        rel = lpr / base_lines[r["scene"]] if base_lines.get(r["scene"]) else float("nan")   # (tiny-mode scenes fetch nothing from global memory in the loops: no lines to count)
        ms = r.get("k_trace_bounce_ms", 0.0)
        relms = ms / ctl_ms[r["scene"]] if ctl_ms.get(r["scene"]) else float("nan")
        out.append(f"{r['scene']:8} {r['order']:74} {r['ch']['node_lines_per_step']:15.2f} {r['ch']['node_lanes']:6.1f} {r['ch']['tri_lines_per_round']:15.2f} {r['ch']['tri_lanes']:6.1f} "
                   f"{lpr:10.2f} {rel:6.2f} {ms:10.3f} {relms:7.2f} {r.get('k_trace_shadow_ms', 0.0):9.3f} {r.get('same', '?'):>5}")
        if r["order"].startswith("i + shadow"):
            out.append("")
    txt = "\n".join(out) + "\n"
    open(os.path.join(ROOT, "gpurun_out", f"{tag}_wave_lines.txt"), "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
