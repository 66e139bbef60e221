#!/usr/bin/env python3
"""scripts/configs.py -- the BASELINE.json configurations on ONE MI355X (run inside gpurun), one bench.py run each:
Mrays/s and ms/frame into profiles/<round>_configs.json.  The frame sizes are BASELINE.json's; where the full sample
count is large (C4: 512 spp, C5: 1024 spp = 2^32 primary samples) fewer timed frames are taken; every configuration runs at
its FULL frame size and sample count.

    python scripts/configs.py r02            # -> profiles/r02_configs.json
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = [
    ("C2", "sphere 512x512x128spp (configs[1])", ["--scene", "sphere"]),
    ("headline", "layered Cornell box 512x512x128spp", []),
    ("headline-refbvh", "the same frame, BVH from the restated reference compiler (leaves of <= 10 triangles)", ["--scene", "cornell-refbvh"]),
    ("C3", "layered Cornell box 1024x1024x256spp (configs[2])", ["--width", "1024", "--height", "1024", "--spp", "256"]),
    ("C4", "material-ball (58,682 tris; stand-in for the Mitsuba scene) 1920x1080x512spp (configs[3])",
     ["--scene", "material-ball", "--width", "1920", "--height", "1080", "--spp", "512", "--steps", "2"]),
    ("C5", "instanced (1,060-tri mesh x 1,024 instances = 1.09 M tris) 2048x2048x1024spp: 2^32 primary samples per frame (configs[4])",
     ["--scene", "instanced", "--width", "2048", "--height", "2048", "--spp", "1024", "--steps", "2"]),
    ("C5-16spp", "the same scene and frame at 16 of its 1024 spp (the line of rounds 1-3)",
     ["--scene", "instanced", "--width", "2048", "--height", "2048", "--spp", "16"]),
    ("terrain", "terrain (1.0 M unique triangles: real HBM gathers) 1024x1024x32spp", ["--scene", "terrain", "--width", "1024", "--height", "1024", "--spp", "32"]),
]


def config_c1():
    """configs[0]: sphere 256x256x16spp "via the reference OpenCL CPU device (plumbing, no GPU)".  There is no OpenCL runtime in the
    image, so its stand-in is the CPU restatement of the same kernels, oracle/polaris_oracle.cpp -- the checker,
    timed here as a baseline and nothing else -- in the reference's own CPU schedule (work-items in ascending order, one
    sample after the other; threads split the work-items of a kernel) and in the sample-parallel mode bench.py's cpu_baseline uses."""
    import time

    sys.path.insert(0, ROOT)
    from polaris_amd.hostinfo import size_openmp

    cores = size_openmp()
    from oracle import pybind as ob
    from polaris_amd import scenes

    W = H = 256
    spp, B = 16, 5
    sc = scenes.SCENES["sphere"](1.0)
    seeds = scenes.make_seeds(spp, B)
    orc = ob.Oracle("oracle")
    res = {}
    for mode, flags in (("reference_order", ob.FIX_EMITTER_INDEX), ("sample_parallel", ob.FIX_EMITTER_INDEX | ob.PARALLEL_SAMPLES)):
        orc.trace(sc, ob.make_request(W, H, spp=1, bounces=B), seeds[: 1 + B], flags=flags)  # warm up (thread team, page faults)
        t = time.perf_counter()
        _, st, _ = orc.trace(sc, ob.make_request(W, H, spp=spp, bounces=B), seeds, flags=flags)
        dt = time.perf_counter() - t
        res[mode] = {"ms_per_frame": round(dt * 1e3, 2), "Mrays_per_s": round(st.total_rays() / dt / 1e6, 2), "rays_per_frame": st.total_rays()}
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        cpu = "unknown CPU"
    return {"what": "sphere 256x256x16spp on the CPU restatement (stand-in for the reference's OpenCL CPU device; plumbing, no GPU) (configs[0])",
            "kind": "port (oracle/polaris_oracle.cpp; the checker, timed as a baseline only)", "cores": cores, "cpu": cpu, **res,
            "workload": f"{sc.name} {W}x{H} {spp}spp {B} bounces rr>=3, {sc.num_triangles} tris"}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    out = {"command": "python bench.py --steps 3 --warmup 1 --no-cpu-baseline <args>", "configs": {}}
    try:
        out["configs"]["C1"] = config_c1()
        print("C1", out["configs"]["C1"]["sample_parallel"], flush=True)
    except Exception as e:  # the GPU lines below do not depend on it
        out["configs"]["C1"] = {"error": str(e)}
    for key, what, args in CONFIGS:
        base = ["--steps", "3", "--warmup", "1"]
        if "--steps" in args:
            base = ["--warmup", "1"]
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *base, "--no-cpu-baseline", "--no-live-counters", *args], capture_output=True, text=True, cwd=ROOT)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            out["configs"][key] = {"what": what, "error": p.stderr[-400:]}
            continue
        d = json.loads(line[-1])
        out["configs"][key] = {"what": what, "args": args, "Mrays_per_s": round(d["value"], 1), "ms_per_frame": round(d["ms_per_frame"], 2),
                               "rays_per_frame": d["config"]["rays_per_frame"], "workload": d["config"]["workload"],
                               "ray_counts": d["config"].get("ray_counts"),   # per bounce and per class, with the occluded fraction of the shadow rays
                               "kernels_isolated_ms_per_frame": d.get("kernels_isolated_ms_per_frame"), "roofline": d.get("roofline"),
                               "roofline_per_kernel": {k: {f: v.get(f) for f in ("timer", "ms_per_frame", "launches", "achieved", "frac")} for k, v in (d.get("roofline_per_kernel") or {}).items()}}
        print(key, out["configs"][key]["Mrays_per_s"], "Mrays/s", out["configs"][key]["ms_per_frame"], "ms/frame", flush=True)
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_configs.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
