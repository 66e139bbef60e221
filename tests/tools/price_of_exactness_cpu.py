#!/usr/bin/env python3
"""tests/tools/price_of_exactness_cpu.py [spp] -- CPU half of VERDICT round 5, item 4 (test infrastructure: it runs the checker).

What would a CONTRACTED build of the path look like against the exact one?  The CPU restatement (oracle/polaris_oracle.cpp) is built a
second time with `-mfma -ffp-contract=fast` -- the compiler may then fuse every a * b + c it finds, which is what a GPU build without
-ffp-contract=off does to the slab tests' and Moeller-Trumbore's products (not the same fusions instruction for instruction: gcc on
x86-64 and clang on gfx950 pick their own; the statistics are what carries over) -- and both trace the headline frame (layered Cornell
box 512 x 512, 5 bounces, RR from bounce 3) with the same seeds.  A fused operation changes a result by an ulp; where that ulp flips a
discrete decision (a hit / miss at a triangle edge, a Russian-roulette or Fresnel branch, a tie between two hits) the path from there on
is ANOTHER path, and the pixel moves by ~1 / spp of a sample's radiance.  Reported: per-pixel RMSE of the mean radiance (the bar of
BASELINE.json is 1e-4), the fraction of pixels that moved by more than accumulated rounding can explain, and the ray counters.

Prints one JSON object; profiles/r06_price_of_exactness.txt quotes it.
"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    from oracle import pybind as ob
    from polaris_amd import scenes
    from polaris_amd.hostinfo import size_openmp

    cores = size_openmp()
    out = os.path.join(ROOT, "oracle", "_build", "libpolaris_oracle_contracted.so")
    flags = ["-std=c++17", "-O2", "-fPIC", "-mfma", "-ffp-contract=fast", "-fno-fast-math", "-fopenmp", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle")]
    subprocess.check_call(["g++", *flags, "-shared", os.path.join(ROOT, "oracle", "polaris_oracle.cpp"), "-o", out])
    ob._PATHS["oracle_contracted"] = (out, "polaris_oracle")
    W = H = 512
    B, rr = 5, 3
    sc = scenes.SCENES["cornell"](W / H)
    seeds = scenes.make_seeds(spp, B)
    res = {}
    for kind in ("oracle", "oracle_contracted"):
        orc = ob.Oracle(kind)
        t = time.perf_counter()
        acc, st, _ = orc.trace(sc, ob.make_request(W, H, spp=spp, bounces=B, rr=rr), seeds, flags=ob.FIX_EMITTER_INDEX | ob.PARALLEL_SAMPLES)
        res[kind] = (acc[..., :3] / spp, st, time.perf_counter() - t)
    a, sa, ta = res["oracle"]
    b, sb, tb = res["oracle_contracted"]
    d = (a - b).astype(np.float64)
    rmse = float(np.sqrt(np.mean(d ** 2)))
    per_pixel = np.sqrt(np.mean(d ** 2, axis=2))
    # rounding alone: every one of the <= spp x (1 + 2 B) contributions to a pixel off by a few ulps -> relative 1e-6 of the pixel at most
    scale = np.maximum(np.sqrt(np.mean(a.astype(np.float64) ** 2, axis=2)), 1e-3)
    moved = per_pixel > 1e-5 * scale
    out = {
        "workload": f"{sc.name} {W}x{H} {spp}spp {B} bounces rr>={rr}", "threads": cores,
        "exact": {"flags": "-O2 -ffp-contract=off (oracle/Makefile)", "seconds": round(ta, 2), "rays": sa.total_rays(),
                  "rays_per_bounce": [int(v) for v in sa.rays_per_bounce[:B]], "occl_per_bounce": [int(v) for v in sa.occl_per_bounce[:B]]},
        "contracted": {"flags": "-O2 -mfma -ffp-contract=fast", "seconds": round(tb, 2), "rays": sb.total_rays(),
                       "rays_per_bounce": [int(v) for v in sb.rays_per_bounce[:B]], "occl_per_bounce": [int(v) for v in sb.occl_per_bounce[:B]]},
        "rmse_mean_radiance": rmse, "bar": 1e-4, "max_abs_pixel_difference": float(np.abs(d).max()), "image_mean": float(a.mean()),
        "pixels_bit_identical": float(np.mean(np.all(a == b, axis=2))),
        "pixels_moved_beyond_rounding": float(np.mean(moved)),
        "ray_counter_difference": sb.total_rays() - sa.total_rays(),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
