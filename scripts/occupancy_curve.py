#!/usr/bin/env python3
"""scripts/occupancy_curve.py <scene> [--pmc] -- the one curve VERDICT round 5 found unmeasured: time of the big-scene traversal
kernels (k_trace<*, 24, 0, false>: closest hit and any hit) against RESIDENT WORKGROUPS PER CU (option trace_wgs_per_cu = 2 .. 6;
one 256-thread workgroup = one wave per SIMD, 24.5 KB of LDS stack: six fit a CU).  Is the 24.5 KB stack that caps a CU at six
workgroups costing anything, or is the CU's gather path (TA) already saturated at four?

The scene is built ONCE (the terrain's numpy tree takes 24 s); every setting traces the same frame with one batch at a time
(overlap = 1: the grid is then exactly CUs x per_cu workgroups) and reads the library's per-kernel HIP-event timers.

    python scripts/occupancy_curve.py terrain            # times -> one JSON line per setting on stdout
    rocprofv3 --kernel-trace --pmc <counters> -d DIR -- python3 scripts/occupancy_curve.py terrain --pmc
                                                          # one frame per setting; scripts/occupancy_pmc.py groups the dispatches of
                                                          # k_trace by GRID SIZE (= CUs x per_cu x 256 threads), i.e. by setting
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {  # scene, W, H, spp (the frames of scripts/pmc_big.sh)
    "terrain": ("terrain", 1024, 1024, 32),
    "C4": ("material-ball", 1920, 1080, 32),
    "C5": ("instanced", 2048, 2048, 16),
}


def main():
    cfg = sys.argv[1]
    pmc = "--pmc" in sys.argv
    settings = [int(v) for v in os.environ.get("PER_CU", "2,3,4,5,6").split(",")]
    from polaris_amd import ctypes_api as T
    from polaris_amd import scenes
    from polaris_amd.tracer import ChangeType, HipTracer, UpdateMode

    scene, W, H, spp = CONFIGS[cfg]
    B, rr = 5, 3
    sc = scenes.SCENES[scene](W / H)
    seeds = scenes.make_seeds(spp, B)
    tr = HipTracer("occ", 0)
    tr.Init()
    for kv in filter(None, os.environ.get("EXTRA_OPTS", "").split(",")):      # e.g. EXTRA_OPTS=trace_spill=1 PER_CU=5,6,7,8 (options that shape the upload)
        k, v = kv.split("=")
        tr.set_option(k, int(v))
    tr.UpdateState(UpdateMode.Synchronous, ChangeType.FrameDimensions, (W, H))
    tr.UpdateState(UpdateMode.Synchronous, ChangeType.SceneData, sc)
    tr.UpdateState(UpdateMode.Synchronous, ChangeType.CameraData, sc)
    req = T.BlockRequest()
    req.frame_w, req.frame_h, req.block_x, req.block_y, req.block_w, req.block_h = W, H, 0, 0, W, H
    req.samples_per_pixel, req.num_bounces, req.min_bounces_for_rr = spp, B, rr
    req.exposure, req.seed = 1.2, 0
    tr.set_option("overlap", 1)
    names = ("intersect", "occlusion", "shade_first", "shade_sort", "shade_plain", "shade_wave", "generate", "fold", "scan", "resolve")
    for per_cu in settings:
        tr.set_option("trace_wgs_per_cu", per_cu)
        tr.set_option("time_kernels", 0)
        frames = 1 if pmc else 3
        if not pmc:
            req.accumulated_samples = 0
            tr.Trace(req, seeds)                      # warm-up (buffers, clocks)
        tr.set_option("time_kernels", 0 if pmc else 1)
        for n in names:
            tr.kernel_ms(n)
        t = time.perf_counter()
        for _ in range(frames):
            req.accumulated_samples = 0
            tr.Trace(req, seeds)
        wall = (time.perf_counter() - t) / frames * 1e3
        st = tr.last_trace_stats
        ms = {n: tr.kernel_ms(n) for n in names}
        out = {"config": cfg, "options": os.environ.get("EXTRA_OPTS", ""), "any_symbol": tr.kernel_symbol("occlusion") if not pmc else None, "scene": sc.name, "frame": [W, H], "spp": spp, "trace_wgs_per_cu": per_cu, "grid_threads": tr.device_cus * per_cu * 256,
               "frame_ms_overlap1": round(wall, 3), "rays": st.total_rays(),
               "closest_hit_ms_per_frame": round(ms["intersect"][0] / frames, 3), "closest_hit_launches": ms["intersect"][1] // frames,
               "any_hit_ms_per_frame": round(ms["occlusion"][0] / frames, 3), "any_hit_launches": ms["occlusion"][1] // frames,
               "shade_ms_per_frame": round(sum(ms[k][0] for k in names if k.startswith("shade")) / frames, 3),
               "closest_symbol": tr.kernel_symbol("intersect") if not pmc else None}
        print(json.dumps(out), flush=True)
    tr.Close()


if __name__ == "__main__":
    main()
