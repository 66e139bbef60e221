#!/usr/bin/env python3
"""scripts/bvh_build_bench.py <tag> -- the device BVH builder (polaris_hip_build_bvh) on the big scenes, inside gpurun:
build time on the device (vertices resident), wall time including the host-side permutation of the triangle arrays, the
time the CPU producers take for the same scene (scenes.py's numpy builder = how the scene was made; polaris_amd/host's
restatement of the reference compiler where it finishes in reasonable time), and what a frame costs on either tree.
-> gpurun_out/<tag>_bvh_build.json"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [("terrain", ["--scene", "terrain", "--width", "1024", "--height", "1024", "--spp", "32"]),
         ("material-ball", ["--scene", "material-ball", "--width", "1920", "--height", "1080", "--spp", "32"]),
         ("instanced", ["--scene", "instanced", "--width", "2048", "--height", "2048", "--spp", "16"]),
         ("cornell", [])]


def bench(args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-kernel-timers", *args],
                       capture_output=True, text=True, cwd=ROOT)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not line:
        return {"error": p.stderr[-300:]}
    d = json.loads(line[-1])
    return {"Mrays_per_s": round(d["value"], 1), "ms_per_frame": round(d["ms_per_frame"], 2), "bvh": d["config"].get("bvh")}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    from polaris_amd import bvh_build, scenes

    out = {}
    for name, args in CASES:
        t = time.perf_counter()
        sc = scenes.SCENES[name]()
        t_scene = time.perf_counter() - t
        builds = []
        for alg in ("sah", "lbvh"):
            bvh_build.rebuild_on_device(sc, max_leaf_tris=4, algorithm=alg)   # (first call: the builder's kernels are loaded)
            for max_leaf in (4, 2):
                t = time.perf_counter()
                new, info = bvh_build.rebuild_on_device(sc, max_leaf_tris=max_leaf, algorithm=alg)
                builds.append({"algorithm": alg, "max_leaf_tris": max_leaf, "device_ms": round(info["device_ms"], 3), "wall_ms_with_readback_and_permutation": round((time.perf_counter() - t) * 1e3, 1),
                               "nodes": info["num_nodes"], "nodes_of_the_cpu_tree": int(len(sc.bvh_nodes))})
        cpu = bench(args)
        out[name] = {"triangles": int(sc.num_triangles), "instances": int(len(sc.mesh_instances)),
                     "cpu_producer_s_whole_scene_numpy_binned_sah": round(t_scene, 2), "device_builds": builds,
                     "frame_on_the_cpu_built_tree": cpu}
        for key, extra in (("sah_leaf4", ["--bvh", "device"]), ("sah_leaf2", ["--bvh", "device", "--bvh-max-leaf", "2"]), ("lbvh_leaf4", ["--bvh", "device-lbvh"])):
            r = bench([*args, *extra])
            if "ms_per_frame" in r and "ms_per_frame" in cpu:
                r["trace_penalty_vs_cpu_tree"] = round(r["ms_per_frame"] / cpu["ms_per_frame"] - 1.0, 3)
            out[name]["frame_on_the_device_built_tree_" + key] = r
        print(name, json.dumps(out[name]), flush=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_bvh_build.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
