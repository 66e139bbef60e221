#!/usr/bin/env python3
"""tests/tools/determinism_stress.py -- trace the same frame many times and require bit-identical accumulators and
counters every time (the shade kernels append rays in whatever order their waves finish: the ORDER is carried as data, so
results must not depend on timing).  Usage (on a GPU box): python tests/tools/determinism_stress.py [frames]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np  # noqa: E402

from polaris_amd.hostinfo import size_openmp  # noqa: E402

size_openmp()


def main():
    from conftest import make_hip_tracer
    from oracle import pybind as ob
    from polaris_amd import scenes

    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    for name, W, H, spp in (("cornell", 512, 512, 128), ("materials", 256, 256, 64), ("cubes", 256, 256, 64), ("material-ball-small", 256, 256, 32)):
        sc = scenes.SCENES[name](W / H)
        seeds = scenes.make_seeds(spp, 5)
        tr = make_hip_tracer(sc, W, H)
        try:
            req = ob.make_request(W, H, spp=spp, bounces=5)
            first = None
            for i in range(frames):
                tr.Trace(req, seeds)
                acc = tr.read_accumulator(0)
                st = tr.last_trace_stats
                sig = (hashlib.sha256(acc.tobytes()).hexdigest(), tuple(st.rays_per_bounce[:5]), tuple(st.occl_per_bounce[:5]), st.shaded_hits, st.unoccluded)
                if first is None:
                    first = sig
                elif sig != first:
                    raise SystemExit(f"{name}: frame {i} differs from frame 0: {sig} vs {first}")
            print(f"{name} {W}x{H}x{spp}spp: {frames} frames bit-identical ({first[0][:16]}..., {sum(first[1]) + sum(first[2])} rays per frame)", flush=True)
        finally:
            tr.Close()


if __name__ == "__main__":
    main()
