#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libpolaris_ref_pm.so).

Runs only where /root/reference exists (this container).  Each fixture holds the inputs (the
compiled scene arrays, request, seeds) and the outputs of the reference's own OpenCL C executed
on the host: the trace accumulator, the ray counters, and the primary-ray taps.  Fixtures are
data; no reference source is stored.

    python tests/tools/make_golden.py [fixture names]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import pybind as ob  # noqa: E402
from polaris_amd import scene_io, scenes  # noqa: E402

CASES = [
    # name, scene key, W, H, spp, bounces, rr, block_y, block_h
    ("cornell_diffuse_32", "cornell-diffuse", 32, 32, 4, 5, 3, 0, 32),
    ("cornell_layered_32", "cornell", 32, 32, 4, 5, 3, 0, 32),
    ("cornell_layered_block", "cornell", 40, 24, 2, 4, 2, 8, 11),   # ragged width, inner row block
    ("sphere_env_32", "sphere", 32, 32, 4, 5, 3, 0, 32),
    ("cubes_instanced_32", "cubes", 32, 32, 4, 5, 3, 0, 32),
    ("materials_32", "materials", 32, 32, 4, 5, 3, 0, 32),
    ("materials_norr_1bounce", "materials", 24, 16, 3, 1, 2, 0, 16),
    ("transformed_instances_32", "transformed", 32, 24, 4, 5, 3, 0, 24),  # rotated / non-uniformly scaled instances, instanced light  # rr disabled: minRR = bounces+1 (cmd/render.go:42-45)
    # a Wavefront scene through the C++ reader + compiler: `instance` statements with rotation and scale (whose boxes the
    # reference's reader computes from the translation alone), mat_expr trees, PNM + PNG textures, a refractive prism
    ("obj_room_40", "obj-room", 40, 30, 4, 5, 3, 0, 30),
    # the C4 stand-in at low tessellation: bump-mapped mix of rough conductor over diffuse, rough glass, textured floor,
    # two area lights + environment, 16:9, an inner row block
    ("material_ball_48", "material-ball-small", 48, 27, 3, 5, 3, 5, 17),
    # the ONE scene file the reference ships: tracer/opencl/fixtures/cube.obj + cube.mtl (a 12-triangle cube, `instance`d twice -- with
    # scale arguments of 0, which types.Scale4 reads as 1 -- one diffuse material, no light, no camera), read where it lies through the
    # C++ reader + compiler.  The harness adds what a render needs and the file lacks -- a camera and an environment light -- in a wrapper
    # scene that `call`s the fixture; only the compiled ARRAYS are stored, never the file's text.
    ("reference_cube_32", "reference-cube", 32, 32, 4, 5, 3, 0, 32),
]

REFERENCE_CUBE = "/root/reference/tracer/opencl/fixtures/cube.obj"


def build_scene(key, aspect):
    if key == "obj-room":
        import tempfile

        sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
        import obj_fixtures
        from polaris_amd import host_api

        return host_api.read_scene(obj_fixtures.write_cornell(tempfile.mkdtemp()), aspect=aspect)
    if key == "reference-cube":
        import tempfile

        from polaris_amd import host_api

        d = tempfile.mkdtemp()
        with open(os.path.join(d, "sky.mtl"), "w") as f:     # the harness's own: an environment light and a background
            f.write("newmtl scene_emissive_material\nKe 1.6 1.5 1.3\nnewmtl scene_diffuse_material\nKd 0.25 0.3 0.4\n")
        with open(os.path.join(d, "wrapper.obj"), "w") as f:
            f.write(f"call {os.path.relpath(REFERENCE_CUBE, d)}\nmtllib sky.mtl\ncamera_fov 0.7\ncamera_eye 1.1 1.3 2.4\ncamera_look -0.5 0 0\ncamera_up 0 1 0\n")
        sc = host_api.read_scene(os.path.join(d, "wrapper.obj"), aspect=aspect)
        sc.name = "reference-cube"
        assert sc.num_triangles == 12 and len(sc.mesh_instances) == 2, "the reference's fixture is a 12-triangle cube instanced twice"
        return sc
    return scenes.SCENES[key](aspect)


def main():
    ob.build_ref()
    ref = ob.Oracle("ref_pm")
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    only = set(sys.argv[1:])  # optional: fixture names to (re)generate
    for name, key, W, H, spp, B, rr, by, bh in CASES:
        if only and name not in only:
            continue
        if key == "reference-cube" and not os.path.exists(REFERENCE_CUBE):
            print(f"{name}: skipped, {REFERENCE_CUBE} is not here (the build container has it)")
            continue
        sc = build_scene(key, W / H)
        seeds = scenes.make_seeds(spp, B, base=0xC0FFEE + len(name))
        req = ob.make_request(W, H, spp=spp, bounces=B, rr=rr, block_y=by, block_h=bh)
        acc, st, taps = ref.trace(sc, req, seeds, tap_sample=0)
        fb = ref.tonemap(acc, 1.0 / spp, 1.2)
        d = scene_io.scene_to_dict(sc)
        d.update(req=np.array([W, H, spp, B, rr, by, bh], dtype=np.int64), seeds=seeds,
                 accum=acc[..., :3].copy(), framebuffer=fb,
                 rays_per_bounce=np.array(list(st.rays_per_bounce[:B]), dtype=np.int64),
                 occl_per_bounce=np.array(list(st.occl_per_bounce[:B]), dtype=np.int64),
                 counters=np.array([st.primary_rays, st.indirect_rays, st.occlusion_rays, st.shaded_hits, st.shaded_misses,
                                    st.unoccluded], dtype=np.int64),
                 primary_rays=taps["primary_rays"], primary_hit=taps["primary_hit"], primary_wuvt=taps["primary_wuvt"],
                 primary_tri=taps["primary_tri"], throughput0=taps["throughput0"][:, :3].copy(),
                 provenance=np.array(ref.describe()))
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **d)
        print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, rays/bounce {list(st.rays_per_bounce[:B])}")


if __name__ == "__main__":
    main()
