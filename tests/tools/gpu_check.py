"""Ad-hoc GPU check: HIP trace vs CPU oracle on every synthetic scene (run through gpurun)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from polaris_amd import scenes
from polaris_amd.tracer import HipTracer, UpdateMode, ChangeType
from oracle import pybind as ob

orc = ob.Oracle("oracle")
W = H = int(os.environ.get("RES", "64")); spp = int(os.environ.get("SPP", "4")); B = 5
for name in ["cornell-diffuse", "cornell", "sphere", "cubes", "materials", "transformed"]:
    sc = scenes.SCENES[name]()
    seeds = scenes.make_seeds(spp, B)
    want, ws, wt = orc.trace(sc, ob.make_request(W, H, spp=spp, bounces=B), seeds, tap_sample=0)
    for exact in (1, 0):
        tr = HipTracer("t", 0); tr.Init()
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.FrameDimensions, (W, H))
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.SceneData, sc)
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.CameraData, sc)
        tr.set_option("exact_accumulate", exact)
        tr.set_option("packet_primary", int(os.environ.get("PACKET", "0")))
        req = ob.make_request(W, H, spp=spp, bounces=B)
        tr.Trace(req, seeds)
        got = tr.read_accumulator(0); gs = tr.last_trace_stats
        taps = tr.tap_primary(ob.make_request(W, H, spp=spp, bounces=B), int(seeds[0]))
        tr.Close()
        biteq = np.array_equal(got[..., :3].view(np.uint32), want[..., :3].view(np.uint32))
        rmse = float(np.sqrt(np.mean((got[..., :3] / spp - want[..., :3] / spp) ** 2)))
        print(f"{name:16s} exact={exact} biteq={biteq} rmse={rmse:.3e} maxabs={np.abs(got[...,:3]-want[...,:3]).max():.3e}")
        print("   rays", list(gs.rays_per_bounce[:B]), list(ws.rays_per_bounce[:B]))
        print("   occl", list(gs.occl_per_bounce[:B]), list(ws.occl_per_bounce[:B]))
        print("   hits", gs.shaded_hits, ws.shaded_hits, "miss", gs.shaded_misses, ws.shaded_misses, "emit", gs.emitter_hits, ws.emitter_hits, "unocc", gs.unoccluded, ws.unoccluded)
        for k in ("primary_rays", "primary_hit", "primary_wuvt", "primary_tri"):
            a, b = taps[k], wt[k]
            if a.dtype == np.float32: a, b = a.view(np.uint32), b.view(np.uint32)
            print("   tap", k, "equal" if np.array_equal(a, b) else f"DIFF {np.sum(a != b)}")
