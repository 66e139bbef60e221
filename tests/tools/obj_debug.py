"""Debug helper: where do HIP and the oracle differ on the OBJ fixture scene?"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import obj_fixtures
from conftest import make_hip_tracer, bits
from oracle import pybind as ob
from polaris_amd import host_api, scenes

W, H, spp = 64, 48, 1
o = ob.Oracle("oracle")
base = obj_fixtures.cornell_obj()
variants = {"full": base,
            "no-prism": "\n".join(l for l in base.split("\n") if not l.startswith("instance prism")),
            "no-cubes": "\n".join(l for l in base.split("\n") if not l.startswith("instance cube")),
            "room-only": "\n".join(l for l in base.split("\n") if not (l.startswith("instance cube") or l.startswith("instance prism"))),
            "all-white": base.replace("usemtl tall", "usemtl white").replace("usemtl glass", "usemtl white").replace("usemtl bumpy", "usemtl white").replace("usemtl floor", "usemtl white")}
for name, text in variants.items():
    d0 = tempfile.mkdtemp()
    obj_fixtures.write_cornell(d0)
    open(os.path.join(d0, "room.obj"), "w").write(text + "\n")
    sc = host_api.read_scene(os.path.join(d0, "room.obj"), aspect=W / H)
    for B in (3, 5):
        req = ob.make_request(W, H, spp=spp, bounces=B, rr=B + 1)
        seeds = scenes.make_seeds(spp, B, base=11)
        want, wst, taps = o.trace(sc, req, seeds, tap_sample=0)
        for ml in (0, 2):
            tr = make_hip_tracer(sc, W, H, exact_accumulate=1, max_leaf_tris=ml)
            tr.Trace(req, seeds)
            got, st = tr.read_accumulator(0), tr.last_trace_stats
            tr.Close()
            d = (bits(got[..., :3]) != bits(want[..., :3])).any(axis=2).reshape(-1)
            print(f"{name} B={B} leaf={ml}: {d.sum()} pixels differ; max {np.abs(got - want).max():.3g}; emitter hits {st.emitter_hits} vs {wst.emitter_hits}; "
                  f"rays {list(st.rays_per_bounce[:B])} vs {list(wst.rays_per_bounce[:B])}")
